#!/usr/bin/env python3
"""A model of the lock-step launch of the cfg4 batch on the CPU: pieces with the evaluations the device counted for them
(tools/dump_piece_costs.py), 512 persistent workgroups of four slots, two per CU, a queue handed out most expensive first -- to
try cuts of the scans without a GPU.  Round of a workgroup: 50.2 k cycles next to its partner, 40.7 k alone on the CU
(profiles/r04_k_phases_mc_wg2.txt).
    python tools/schedule_sim.py gpurun_out/piece_costs.npz"""
import heapq, sys
import numpy as np

d = np.load(sys.argv[1])
kinds, alphas = d['kinds'], d['alphas']
NA, NS = 100, len(kinds)
warm = d['evals_split1'].astype(float)                 # [scan][alpha]: every alpha warm from its neighbour (alpha 0: the cold start at the top)
# cold start at position a: the first alpha of a piece that begins there, from the cuts that have one
cold = np.full((NS, NA), np.nan)
after = np.full((NS, NA), np.nan)                      # the alpha right behind a cold start, against its warm cost
for split in (50, 33, 25, 20, 15, 10, 4):
    ev = d['evals_split%d' % split].astype(float)
    for s in range(split):
        a0 = NA * s // split
        m = np.isnan(cold[:, a0])
        cold[m, a0] = ev[m, a0]
        if a0 + 1 < NA * (s + 1) // split:
            m = np.isnan(after[:, a0 + 1])
            after[m, a0 + 1] = ev[m, a0 + 1] - warm[m, a0 + 1]
for s in range(NS):                                    # positions without data: the nearest one with
    have = np.flatnonzero(~np.isnan(cold[s]))
    for a in range(NA):
        if np.isnan(cold[s, a]):
            cold[s, a] = cold[s, have[np.argmin(np.abs(have - a))]]
    have = np.flatnonzero(~np.isnan(after[s]))
    for a in range(NA):
        if np.isnan(after[s, a]):
            after[s, a] = after[s, have[np.argmin(np.abs(have - a))]] if len(have) else 0.0
normal = kinds == 0


def piece_cost(s, a0, a1):
    c = cold[s, a0] + warm[s, a0 + 1:a1].sum()
    if a1 > a0 + 1:
        c += after[s, a0 + 1]
    return c


def host_estimate(s, a0, a1):
    # the queue order of mxe_chains_upload
    return (a1 - a0) * (4.0 if normal[s] else 3.0) + (16.0 if normal[s] else 6.0) - 1e-3 * np.log10(alphas[a1 - 1])


def simulate(cuts_of, n_wg=512, n_solo=4, both=50.2e3, alone=40.7e3, order='host', verbose=False):
    pieces = []
    for s in range(NS):
        cuts = cuts_of(s)
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            pieces.append((s, a0, a1, piece_cost(s, a0, a1), host_estimate(s, a0, a1)))
    key = (lambda p: -p[4]) if order == 'host' else (lambda p: -p[3])
    pieces.sort(key=key)                      # (stable: python's sort)
    q = [int(round(p[3])) for p in pieces]
    nq, head = len(q), 0
    slots = np.zeros((n_wg, 4), dtype=int)
    for b in range(n_wg):                     # first pieces: slot by slot from the head of the queue
        for k in range(4):
            if head < nq:
                slots[b, k] = q[head]; head += 1
    partner = lambda b: (b + n_wg // 2) % n_wg
    active = np.ones(n_wg, dtype=bool)
    for b in range(n_solo):                   # the partners of the workgroups with the longest pieces leave at once
        active[partner(b)] = False
        for k in range(4):                    # (their pieces go back: in the kernel they never take any)
            pass
    rounds = np.zeros(n_wg, dtype=int)
    t_end = np.zeros(n_wg)
    heap = []
    for b in range(n_wg):
        if active[b]:
            heapq.heappush(heap, ((both if active[partner(b)] else alone), b))
    # (workgroups that left at once: their first pieces were never taken -- redo the hand-out without them)
    if n_solo:
        head = 0
        slots[:] = 0
        for b in range(n_wg):
            if active[b]:
                for k in range(4):
                    if head < nq:
                        slots[b, k] = q[head]; head += 1
    total_rounds = 0
    while heap:
        t, b = heapq.heappop(heap)
        rounds[b] += 1
        total_rounds += 1
        for k in range(4):
            if slots[b, k] > 0:
                slots[b, k] -= 1
                if slots[b, k] == 0 and head < nq:
                    slots[b, k] = q[head]; head += 1
        if slots[b].any():
            heapq.heappush(heap, (t + (both if active[partner(b)] else alone), b))
        else:
            active[b] = False
            t_end[b] = t
    r = rounds[rounds > 0]
    res = dict(pieces=nq, evals=sum(q), makespan_cycles=t_end.max(), mean_cycles=t_end[t_end > 0].mean(), rounds_mean=r.mean(), rounds_max=r.max(),
               wg_rounds=total_rounds, ms=t_end.max() / 2.4e6)
    return res


def uniform(n):
    return lambda s: [NA * i // n for i in range(n + 1)]


def show(tag, res):
    print('%-58s pieces %5d evals %6d | rounds/wg mean %.1f max %d | makespan %.3f Mcycles = %.3f ms at 2.4 GHz (mean wg %.3f)' % (
        tag, res['pieces'], res['evals'], res['rounds_mean'], res['rounds_max'], res['makespan_cycles'] / 1e6, res['ms'], res['mean_cycles'] / 1e6))


if __name__ == '__main__':
    print('measured: rounds per workgroup mean 33.5 max 45 (solo), slowest 1.91 Mcycles, mean 1.683 Mcycles, 17176 workgroup-rounds, 65159 evaluations')
    for n in (12, 13, 14, 15, 16, 17, 18):
        show('uniform %d pieces per scan, host order' % n, simulate(uniform(n)))
    show('uniform 15, order by true cost', simulate(uniform(15), order='true'))
    show('uniform 15, no solo workgroups', simulate(uniform(15), n_solo=0))

    # ---- a-priori cost models for the queue order ----
    mean_cold = {k: cold[kinds == k].mean(axis=0) for k in (0, 1)}
    mean_warm = {k: warm[kinds == k].mean(axis=0) for k in (0, 1)}
    mean_after = {k: after[kinds == k].mean(axis=0) for k in (0, 1)}

    def model_mean(s, a0, a1):
        k = kinds[s]
        return mean_cold[k][a0] + mean_warm[k][a0 + 1:a1].sum() + (mean_after[k][a0 + 1] if a1 > a0 + 1 else 0.0)

    def model_formula(s, a0, a1):
        # what mxe_chains_upload can know: kind and position in the logarithmic range
        x = lambda a: a / (NA - 1.0)
        if normal[s]:
            return 9.5 + 13.0 * max(0.0, x(a0) - 0.3) + sum(2.6 + 1.0 * x(a) for a in range(a0 + 1, a1))
        return 4.5 + sum(2.0 + 1.0 * x(a) for a in range(a0 + 1, a1))
    print('mean evaluations by position, plus-minus: cold', np.round(mean_cold[1][::10], 1), 'warm', np.round(mean_warm[1][5::10], 2))
    print('mean evaluations by position, normal:     cold', np.round(mean_cold[0][::10], 1), 'warm', np.round(mean_warm[0][5::10], 2))
    import types
    for name, fn in (('mean-by-position model', model_mean), ('formula model', model_formula)):
        g = globals()
        keep = g['host_estimate']
        g['host_estimate'] = fn
        for n in (14, 15, 16):
            show('uniform %d, order by the %s' % (n, name), simulate(uniform(n)))
        show('uniform 15, %s, no solo' % name, simulate(uniform(15), n_solo=0))
        g['host_estimate'] = keep
