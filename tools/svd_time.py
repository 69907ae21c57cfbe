"""device time of one mxe_kernel_svd call (cfg2 mesh); also usable with experiment builds whose
status is not OK (prints the time regardless)"""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
from maxent_amd import device, synthetic
lib = device.load_library()
tau, omega = synthetic.grids(200, 500)
w = np.ascontiguousarray(np.asarray(omega), dtype=float); d = np.ascontiguousarray(omega.delta); t = np.ascontiguousarray(tau)
b = np.zeros(1); U = np.empty((1, 200, 128)); S = np.empty((1, 128)); V = np.empty((1, 500, 128))
ns = np.zeros(1, dtype=np.int32); info = np.zeros((1, 3), dtype=np.int32); ms = ctypes.c_float(0)
P = device._p
for i in range(3):
    rc = lib.mxe_kernel_svd(0, 200, 500, P(t), P(w), P(d), 40.0, 1, P(b), 1e-14, 128, None, P(U), P(S), P(V), P(ns), P(info), ctypes.byref(ms))
    print('rc', rc, 'ms %.3f' % ms.value, 'n_s', ns[0], 'qr_rank, sweeps, status', info[0])
