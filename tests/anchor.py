"""Reference-anchored truth for the GPU parity tests: the oracle port of the reference runs the alpha scan,
and ITS iterates (not the GPU's) are polished in extended precision (oracle/hp_truth.py).  TEST
INFRASTRUCTURE: imports oracle/."""
import numpy as np

from oracle import ref_numpy as R, hp_truth


def truth(p, alpha_scaled, v_start, entropy, iters=6):
    """H of the extended-precision fixed point reached from ``v_start``; the polish must have CONVERGED and be finite
    (a NaN truth used to pass silently through ``max(worst, nan)``: VERDICT r04)"""
    info = {}
    _, Ht = hp_truth.polish(p.K, p.G, p.err, p.D, p.V, p.S, alpha_scaled, v_start, entropy, iters=iters, info=info)
    assert info['converged'], ('the extended-precision polish did not converge', info)
    assert np.all(np.isfinite(Ht)), 'truth is not finite'
    return Ht


def rel_l2_checked(H, Ht):
    """relative L2 distance of H from the truth Ht; never NaN (asserted)"""
    e = np.linalg.norm(np.asarray(H) - Ht) / np.linalg.norm(Ht)
    assert np.isfinite(e), 'distance from the truth is not finite'
    return e


def truth_rows(p, delta, alphas_scaled, n_data, rows, entropy):
    """{alpha index: H_truth} for the rows asked for; ``p``: R.Problem; ``alphas_scaled``: alpha * n_data"""
    ref = R.alpha_loop(p, delta, np.asarray(alphas_scaled) / n_data)
    out = {}
    for ia in rows:
        out[ia] = truth(p, alphas_scaled[ia], ref['v'][ia], entropy, iters=6)
    return out, ref
