#!/usr/bin/env python3
"""Cycles per round of chain_kernel_lv by phase, from the stamp build (make -C maxent_amd/csrc prof):
    MAXENT_AMD_LIB=maxent_amd/lib/libmaxent_hip_prof.so python tools/lv_phases.py [cfg2|cfg3|shard8]
Stamps per wave: 0 accept, 1 solve + step, 2 wait at the barrier behind the home section, 3 row pass, 4 fused pass,
5 wait at the barrier behind the passes; 7 rounds."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MAXENT_AMD_LIB', os.path.join(ROOT, 'maxent_amd', 'lib', 'libmaxent_hip_prof.so'))
sys.argv = sys.argv[:1] + ['--cases'] + (sys.argv[1:] or ['cfg2'])
import bench
from maxent_amd import device, synthetic

def case(name):
    if name == 'cfg2':
        batch = bench.build_batch(2, 200, 500, 100, 0)
        _, _, _, G1 = synthetic.single_G(200, 500)
        batch['Gmat'] = G1[None, None, :]
        batch['elems'], batch['kinds'], batch['v0'] = [(0, 0)], batch['kinds'][:1], batch['v0'][:1]
        return batch, [0]
    if name == 'cfg3':
        return bench.build_batch(4, 200, 500, 100, 0), list(range(16))
    N = int(name[5:])
    return bench.build_batch(16, 200, 500, 100, 0), [e for e in range(256) if e % N == N - 1]

lib = device.load_library()
lib.mxe_prof_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
for name in sys.argv[2:]:
    batch, which = case(name)
    for lab, opts in (('first pass of two (tol 1e-5)', dict()), ('binary32 launch', dict(precision=device.PRECISION_F32))):
        ctx = bench.stage(batch, 0, which)
        ctx.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which], device.default_opts(**opts))
        for _ in range(2):
            ctx.launch(); ctx.sync()
        info = ctx.last_launch_info()
        n_sub = len(which) * 100
        prof = np.zeros((n_sub + 8192, 8), dtype=np.int64)
        assert lib.mxe_prof_fetch(ctx._h, prof.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
        n_wg = info['n_workgroups']
        pw = prof[:n_wg * 8].reshape(n_wg, 8, 8).astype(float)
        rounds = pw[:, 0, 7]
        print('== %s, %s: %s, %d workgroups, kernel(s) %.3f ms, rounds per workgroup mean %.1f max %.0f' %
              (name, lab, info['kernel'], n_wg, ctx.last_kernel_ms(), rounds.mean(), rounds.max()))
        names = ['accept', 'solve+step', 'wait: home', 'row pass', 'fused pass', 'wait: passes']
        for cls, sl in (('home waves 0-3', slice(0, 4)), ('partner waves 4-7', slice(4, 8))):
            tot = pw[:, sl, :6].sum(axis=(0, 1)) / (rounds.sum() * 4)
            print('  %-18s ' % cls + '  '.join('%s %.0f' % (n, c) for n, c in zip(names, tot)) + '   sum %.0f cycles per round' % tot.sum())
        w = int(np.argmax(rounds))
        print('  deepest workgroup %d: %d rounds, home wave 0: ' % (w, rounds[w]) + '  '.join('%s %.0f' % (n, c / rounds[w]) for n, c in zip(names, pw[w, 0, :6])))
        ctx.close()
