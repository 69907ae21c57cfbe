"""tail stealing in the model of tools/schedule_sim.py: a slot that finds the queue empty takes the lower half of the not yet
started alphas of the piece with the most of them (cold start there)."""
import heapq, sys, types
import numpy as np
sys.argv = [sys.argv[0], sys.argv[1]]
src = open('tools/schedule_sim.py').read().split("if __name__ == '__main__':")[0]
m = types.ModuleType('sim'); exec(compile(src, 'sim', 'exec'), m.__dict__)
cold, warm, after, NA, NS = m.cold, m.warm, m.after, m.NA, m.NS

def alpha_cost(s, a0, a):
    if a == a0: return int(round(cold[s, a]))
    c = warm[s, a] + (after[s, a] if a == a0 + 1 else 0.0)
    return max(1, int(round(c)))

def simulate(n_pieces=15, n_wg=512, n_solo=4, both=50.2e3, alone=40.7e3, steal_min=0, order='host'):
    pieces = []
    for s in range(NS):
        cuts = [NA * i // n_pieces for i in range(n_pieces + 1)]
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            pieces.append((s, a0, a1, m.piece_cost(s, a0, a1), m.host_estimate(s, a0, a1)))
    pieces.sort(key=(lambda p: -p[4]) if order == 'host' else (lambda p: -p[3]))
    head = 0
    partner = lambda b: (b + n_wg // 2) % n_wg
    active = np.ones(n_wg, dtype=bool)
    for b in range(n_solo): active[partner(b)] = False
    # slot state: [s, a0, a, a1, left]
    slots = [[None] * 4 for _ in range(n_wg)]
    def take(b, k):
        nonlocal head
        if head < len(pieces):
            s, a0, a1 = pieces[head][:3]; head += 1
            slots[b][k] = [s, a0, a0, a1, alpha_cost(s, a0, a0)]
            return True
        if steal_min:
            best, bb = 0, None
            for b2 in range(n_wg):
                for k2 in range(4):
                    p = slots[b2][k2]
                    if p is not None:
                        rem = p[3] - p[2] - 1
                        if rem > best: best, bb = rem, p
            if bb is not None and best >= steal_min:
                give = best // 2 if best > 1 else 1
                give = max(give, 1)
                new_end = bb[3] - give
                slots[b][k] = [bb[0], new_end, new_end, bb[3], alpha_cost(bb[0], new_end, new_end)]
                bb[3] = new_end
                return True
        slots[b][k] = None
        return False
    for b in range(n_wg):
        if active[b]:
            for k in range(4): take(b, k)
    heap = [((both if active[partner(b)] else alone), b) for b in range(n_wg) if active[b]]
    heapq.heapify(heap)
    rounds = np.zeros(n_wg, dtype=int); t_end = np.zeros(n_wg); evals = 0
    while heap:
        t, b = heapq.heappop(heap)
        rounds[b] += 1
        for k in range(4):
            p = slots[b][k]
            if p is None:
                if steal_min: take(b, k)      # an idle slot looks again every round
                continue
            p[4] -= 1; evals += 1
            if p[4] == 0:
                p[2] += 1
                if p[2] >= p[3]: take(b, k)
                else: p[4] = alpha_cost(p[0], p[1], p[2])
        if any(p is not None for p in slots[b]):
            heapq.heappush(heap, (t + (both if active[partner(b)] else alone), b))
        else:
            active[b] = False; t_end[b] = t
    r = rounds[rounds > 0]
    return dict(ms=t_end.max() / 2.4e6, evals=evals, rmean=r.mean(), rmax=r.max(), mean_ms=t_end[t_end > 0].mean() / 2.4e6)

for sm in (0, 8, 6, 4, 3, 2):
    for npc in (15, 12, 10):
        r = simulate(npc, steal_min=sm)
        print('pieces %2d steal_min %d: %.3f ms, evals %d, rounds mean %.1f max %d, mean wg %.3f ms' % (npc, sm, r['ms'], r['evals'], r['rmean'], r['rmax'], r['mean_ms']))
