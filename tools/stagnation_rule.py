#!/usr/bin/env python3
"""A stagnation rule for the finishing pass, tried offline on recorded histories (diagnostic build -DMXE_DEBUG_HIST: the stopping
quantity of every finished alpha at iterations 10, 20, 40, 60, 80, 100, 150, 200, 300, ... 900): stop an alpha at iteration n
(a multiple of 100, n >= start) when q(n) > thr * q(n - span) and q(n) > floor.  How many alphas that DID converge would it have
stopped, how many iterations of the ones that did not does it save.
    python tools/stagnation_rule.py gpurun_out/hist_s7.txt gpurun_out/hist_s11.txt ..."""
import sys
import numpy as np
marks = [10, 20, 40, 60, 80, 100, 150, 200, 300, 400, 500, 600, 700, 800, 900]
sets = {}
for path in sys.argv[1:]:
    rows = []
    for l in open(path):
        p = l.split('|'); h = p[0].split()
        rows.append((int(h[1]), int(h[2]), np.array(p[1].split(), float)[:15]))
    sets[path] = rows


def rule(rows, thr, span, start, floor):
    killed, saved, total = [], 0, 0
    for cv, ni, a in rows:
        stop = None
        for k, m in enumerate(marks):
            if m < start or m > ni or (m - span) not in marks:
                continue
            q, q0 = a[k], a[marks.index(m - span)]
            if q0 > 0 and q > thr * q0 and q > floor:
                stop = m
                break
        if cv == 0:
            total += ni
        if stop is not None:
            if cv:
                killed.append((ni, stop))
            else:
                saved += ni - stop
    return killed, saved, total


for path, rows in sets.items():
    n_conv = sum(1 for r in rows if r[0]); n_un = len(rows) - n_conv
    print('%s: %d finished alphas, %d converged (%d of them after more than 300 iterations), %d not' % (
        path, len(rows), n_conv, sum(1 for r in rows if r[0] and r[1] > 300), n_un))
for thr in (0.7, 0.75, 0.8, 0.85, 0.9):
    for span, start in ((300, 400), (300, 500), (200, 400)):
        for floor in (1e-2, 0.5):
            out = []
            for path, rows in sets.items():
                k, s, t = rule(rows, thr, span, start, floor)
                out.append('%d lost, %d of %d saved' % (len(k), s, t))
            print('thr %.2f span %d start %d floor %.0e: %s' % (thr, span, start, floor, ' | '.join(out)))
