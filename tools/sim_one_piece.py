"""ONE piece per slot, pieces of equal a-priori cost (formula of mxe_chains_upload's cost model), in the model of tools/schedule_sim.py"""
import sys, types
import numpy as np
sys.argv = [sys.argv[0], sys.argv[1]]
src = open('tools/schedule_sim.py').read().split("if __name__ == '__main__':")[0]
m = types.ModuleType('sim'); exec(compile(src, 'sim', 'exec'), m.__dict__)
NA, NS, normal = m.NA, m.NS, m.normal
x = lambda a: a / (NA - 1.0)
def cold_f(s, a): return 9.5 + 13.0 * max(0.0, x(a) - 0.3) if normal[s] else 4.5
def warm_f(s, a): return (2.6 + 1.0 * x(a)) if normal[s] else (2.0 + 1.0 * max(0.0, x(a) - 0.7) * 2.5)
def cuts_for(s, T):
    cuts = [0]; a = 0
    while a < NA:
        c = cold_f(s, a); b = a + 1
        while b < NA and c + warm_f(s, b) <= T: c += warm_f(s, b); b += 1
        cuts.append(b); a = b
    return cuts
def model(s, a0, a1): return cold_f(s, a0) + sum(warm_f(s, a) for a in range(a0 + 1, a1))
m.host_estimate = model
for T in (20, 24, 26, 28, 29, 30, 31, 32, 34, 36, 40):
    for Tn in (T, T + 4, T + 8):
        cf = lambda s, T=T, Tn=Tn: cuts_for(s, Tn if normal[s] else T)
        npieces = sum(len(cf(s)) - 1 for s in range(NS))
        for ns in (0, 4):
            r = m.simulate(cf, n_solo=ns, order='host')
            rt = m.simulate(cf, n_solo=ns, order='true')
            print('T pm %2d normal %2d: pieces %4d evals %5d | solo %d host order %.3f ms (rounds mean %.1f max %d) | true order %.3f ms' % (
                T, Tn, npieces, r['evals'], ns, r['ms'], r['rounds_mean'], r['rounds_max'], rt['ms']))
