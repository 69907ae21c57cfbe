"""Reference-anchored truth for the GPU parity tests: the oracle port of the reference runs the alpha scan,
and ITS iterates (not the GPU's) are polished in extended precision (oracle/hp_truth.py).  TEST
INFRASTRUCTURE: imports oracle/."""
import numpy as np

from oracle import ref_numpy as R, hp_truth


def truth_rows(p, delta, alphas_scaled, n_data, rows, entropy):
    """{alpha index: H_truth} for the rows asked for; ``p``: R.Problem; ``alphas_scaled``: alpha * n_data"""
    ref = R.alpha_loop(p, delta, np.asarray(alphas_scaled) / n_data)
    out = {}
    for ia in rows:
        _, Ht = hp_truth.polish(p.K, p.G, p.err, p.D, p.V, p.S, alphas_scaled[ia], ref['v'][ia], entropy, iters=6)
        out[ia] = Ht
    return out, ref
