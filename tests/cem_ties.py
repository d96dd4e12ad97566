"""Accounting for rank ties inside the CEM elite mask (reference ``mpc_controller.py:100-104``).

The reference's "elites" are ``((-returns).argsort(-1) < k).T``: position p of the mask is set iff the candidate
RANKED p-th has an index below k.  The mask therefore depends on the complete ranking, and two candidates whose
returns differ by less than the fp32 error of a rollout may swap ranks and flip two mask positions.  Order
statistics are 1-Lipschitz in the sup norm: if every return is within ``eps`` of the reference's, the candidate a
ranked p-th here and the candidate b ranked p-th by the reference satisfy ``|ref[a] - ref[b]| <= 2 eps``.  So every
legitimate flip has a WITNESS pair (a, b) that close together in the reference's own returns - and a ranking bug
(wrong comparison, wrong tie rule, lost candidate) shows up as a flip without one.
"""

import numpy as np


def rank_flips(got, ref, k):
    """``got``, ``ref``: returns ``[m, n]``.  One record per rank position whose candidate differs between the two
    rankings: ``(env, position, a, b, gap, scale, in_mask)`` with ``gap = |ref[a] - ref[b]|``,
    ``scale = max(1, |ref[a]|, |ref[b]|)`` and ``in_mask`` = the elite mask differs at that position."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    out = []
    for i in range(ref.shape[0]):
        og = (-got[i]).argsort(axis=-1)
        orf = (-ref[i]).argsort(axis=-1)
        for p in np.nonzero(og != orf)[0]:
            a, b = int(og[p]), int(orf[p])
            out.append((i, int(p), a, b, abs(ref[i, a] - ref[i, b]), max(1.0, abs(ref[i, a]), abs(ref[i, b])),
                        bool((a < k) != (b < k))))
    return out


def assert_flips_are_ties(got, ref, k, rtol):
    """Every rank swap (hence every elite-mask flip) must be witnessed by a pair of reference returns closer than
    twice the error bar.  Returns ``(swaps, mask_flips, worst_gap_over_scale)``."""
    flips = rank_flips(got, ref, k)
    worst = 0.0
    for env, pos, a, b, gap, scale, in_mask in flips:
        worst = max(worst, gap / scale)
        assert gap <= 2.0 * rtol * scale, (
            "rank %d of env %d: candidates %d / %d swapped although their reference returns differ by %.3e "
            "(allowed %.3e) - not a tie" % (pos, env, a, b, gap, 2.0 * rtol * scale))
    return len(flips), sum(1 for f in flips if f[6]), worst


def reference_refit(mean, clipped, returns, k, alpha):
    """``mpc_controller.py:101-104`` on float64 arrays: ``clipped [n, m, D]``, ``returns [m, n]``."""
    mask = ((-np.asarray(returns, dtype=np.float64)).argsort(axis=-1) < k).T
    elites = clipped[mask]
    return mean * alpha + (1 - alpha) * np.mean(elites, axis=0), np.std(elites, axis=0)
