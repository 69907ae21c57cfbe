#!/usr/bin/env python3
"""The region of bench.py with batches in flight as rocprofv3 --kernel-trace recorded it: every dispatch of the chain kernel with its
start and end, how many run side by side, how long one takes under that contention, and the rate of the region.
    python tools/in_flight_trace.py <dir of the rocprofv3 run> [n_workgroups of the cut in flight, default 256]
The launches of the cut for batches in flight have 256 workgroups (one batch at a time: 512): that tells the regions apart."""
import csv, glob, os, sys
import numpy as np
root = sys.argv[1]
wg_fl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rows = []
for path in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'chain_kernel_mc' not in r.get('Kernel_Name', ''):
                continue
            gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
            wx = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 256)) or 256)
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), gx // max(wx, 1), int(r.get('Queue_Id', 0) or 0)))
rows.sort()
if not rows:
    sys.exit('no dispatch of chain_kernel_mc in %s' % root)
a = np.array(rows, dtype=np.int64)
print('# dispatches of mxe::chain_kernel_mc: %d; workgroups per dispatch: %s' % (len(a), dict(zip(*np.unique(a[:, 2], return_counts=True)))))
for tag, sel in (('one batch at a time (launches of %d workgroups)' % 512, a[:, 2] != wg_fl), ('batches in flight (launches of %d workgroups)' % wg_fl, a[:, 2] == wg_fl)):
    d = a[sel]
    if len(d) < 8:
        print('# %s: %d dispatches -- too few' % (tag, len(d)))
        continue
    dur = (d[:, 1] - d[:, 0]) * 1e-6
    # side by side: at the start of every dispatch, how many others are still running
    ends = np.sort(d[:, 1])
    running = np.array([np.sum((d[:, 0] <= s) & (d[:, 1] > s)) for s in d[:, 0]])
    # the timed regions are runs of dispatches enqueued back to back (no host synchronisation inside); warm-up and the settling
    # passes are bursts of one launch per context with a wait behind each.  The densest window of W consecutive dispatches (by
    # their ends) is therefore inside a timed region: its rate, and the dispatches in it
    order = np.argsort(d[:, 1])
    d = d[order]; dur = dur[order]; running = running[order]
    W = min(32, len(d) - 1)
    spans = d[W:, 1] - d[:-W, 1]
    i0 = int(np.argmin(spans))
    seg = d[i0:i0 + W + 1]
    print('%s:\n  dispatches %d, queues %d; duration of one dispatch: mean %.4f ms, median %.4f, min %.4f, max %.4f\n'
          '  running side by side at the start of a dispatch (itself included): mean %.2f, max %d\n'
          '  densest window of %d consecutive dispatches (by their ends): %.4f ms per dispatch; in that window one dispatch takes %.4f ms '
          '(mean; min %.4f, max %.4f) and %.2f run side by side'
          % (tag, len(d), len(set(d[:, 3])), dur.mean(), np.median(dur), dur.min(), dur.max(), running.mean(), running.max(),
             W, spans[i0] * 1e-6 / W, dur[i0:i0 + W + 1].mean(), dur[i0:i0 + W + 1].min(), dur[i0:i0 + W + 1].max(),
             running[i0:i0 + W + 1].mean()))
    print('  the first dispatches of that window (start us, end us, duration ms, queue):')
    t0 = seg[:, 0].min()
    for r in seg[np.argsort(seg[:, 0])][:16]:
        print('    %10.1f %10.1f %8.4f  %d' % ((r[0] - t0) * 1e-3, (r[1] - t0) * 1e-3, (r[1] - r[0]) * 1e-6, r[3]))
