"""partners of the workgroups that hold the long normal-entropy pieces stop taking pieces after R rounds (they leave early, the heavy
workgroup runs alone on its CU for its last rounds): in the model of tools/schedule_sim.py"""
import heapq, sys, types
import numpy as np
sys.argv = [sys.argv[0], sys.argv[1]]
src = open('tools/schedule_sim.py').read().split("if __name__ == '__main__':")[0]
m = types.ModuleType('sim'); exec(compile(src, 'sim', 'exec'), m.__dict__)
NA, NS = m.NA, m.NS

def simulate(n_heavy=60, R=20, n_wg=512, n_solo=4, both=50.2e3, alone=40.7e3):
    pieces = []
    for s in range(NS):
        cuts = [NA * i // 15 for i in range(16)]
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            pieces.append((s, a0, a1, m.piece_cost(s, a0, a1), m.host_estimate(s, a0, a1)))
    pieces.sort(key=lambda p: -p[4])
    q = [int(round(p[3])) for p in pieces]
    nq, head = len(q), 0
    half = n_wg // 2
    partner = lambda b: (b + half) % n_wg
    active = np.ones(n_wg, dtype=bool)
    for b in range(n_solo): active[partner(b)] = False
    slots = np.zeros((n_wg, 4), dtype=int)
    for b in range(n_wg):
        if active[b]:
            for k in range(4):
                if head < nq: slots[b, k] = q[head]; head += 1
    limited = set(range(half + n_solo, half + n_heavy))
    rounds = np.zeros(n_wg, dtype=int); t_end = np.zeros(n_wg)
    heap = [((both if active[partner(b)] else alone), b) for b in range(n_wg) if active[b]]
    heapq.heapify(heap)
    while heap:
        t, b = heapq.heappop(heap)
        rounds[b] += 1
        for k in range(4):
            if slots[b, k] > 0:
                slots[b, k] -= 1
                if slots[b, k] == 0 and head < nq and not (b in limited and rounds[b] >= R):
                    slots[b, k] = q[head]; head += 1
        if slots[b].any():
            heapq.heappush(heap, (t + (both if active[partner(b)] else alone), b))
        else:
            active[b] = False; t_end[b] = t
    if head < nq: return None
    r = rounds[rounds > 0]
    return t_end.max() / 2.4e6, r.mean(), r.max()

print('base', simulate(4, 0))
for nh in (20, 40, 60, 80, 100):
    for R in (0, 5, 10, 15, 20, 25):
        print(nh, R, simulate(nh, R))
