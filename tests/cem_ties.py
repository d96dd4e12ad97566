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


def verify_tail_from_product(rollout, low, high, n, m, h, alpha, k, seed_or_state, trace, it0, best_index, chosen, rtol):
    """After a PROVEN rank tie inside the elite mask at iteration ``it0`` the product and the reference legitimately part ways -
    but what the product does from there on is still checkable: the oracle CONTINUED FROM THE PRODUCT'S OWN statistics
    (``trace[it0]['mean' / 'std']``) with the same normal draws (the stream does not depend on the values).  For every later
    iteration: the product's returns against the oracle's (``rtol``), every rank swap a witnessed tie, the refit = the reference's
    arithmetic (``:101-104``) on the product's own returns (1e-9); at the end the chosen index must be the oracle's arg-max up
    to such a tie and the chosen action the float64 first action of THAT candidate, bit for bit.
    ``rollout(seq [h, n * m, act_dim]) -> returns [m, n]`` is the oracle's; ``seed_or_state``: the seed the plan started from (int)
    or the generator state right before its first draw.  Returns the number of later iterations that flipped a mask again."""
    low = np.asarray(low, dtype=np.float64)
    high = np.asarray(high, dtype=np.float64)
    act_dim = low.shape[0]
    D = h * act_dim
    clip_low, clip_high = np.concatenate([low] * h), np.concatenate([high] * h)
    keep = np.random.get_state()
    try:
        if isinstance(seed_or_state, (int, np.integer)):
            np.random.seed(int(seed_or_state))
        else:
            np.random.set_state(seed_or_state)
        for _ in range(it0 + 1):
            np.random.normal(size=(n, m, D))                       # the draws consumed up to and including iteration it0
        mean, std = np.array(trace[it0]["mean"]), np.array(trace[it0]["std"])
        again = 0
        returns_o = first = None
        for it in range(it0 + 1, len(trace)):
            z = np.random.normal(size=(n, m, D))                    # :85
            raw = mean + z * std
            clipped = np.clip(raw, clip_low, clip_high)
            seq = np.transpose(raw.reshape((n * m, h, act_dim)), (1, 0, 2))
            first = seq[0].reshape((m, n, -1))                      # :94
            returns_o = np.asarray(rollout(seq), dtype=np.float64).reshape(m, n)
            got = np.asarray(trace[it]["returns"], dtype=np.float64)
            err = float(np.max(np.abs(got - returns_o) / np.maximum(1.0, np.abs(returns_o))))
            assert err < rtol, "iteration %d continued from the product's statistics: returns off by %.2e" % (it, err)
            _, flips, _ = assert_flips_are_ties(got, returns_o, k, rtol)
            again += 1 if flips else 0
            want_mean, want_std = reference_refit(mean, clipped, got, k, alpha)
            np.testing.assert_allclose(np.broadcast_to(trace[it]["mean"], want_mean.shape), want_mean, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(trace[it]["std"], want_std, rtol=1e-9, atol=1e-12)
            mean, std = np.array(trace[it]["mean"]), np.array(trace[it]["std"])
        if returns_o is not None:                                    # (a flip in the LAST iteration leaves nothing to continue)
            for i in range(m):
                b = int(best_index[i])
                gap = float(returns_o[i].max() - returns_o[i, b])
                assert gap <= 2.0 * rtol * max(1.0, abs(float(returns_o[i].max()))), (
                    "env %d: the chosen candidate %d is %.3e below the oracle's best - not a tie" % (i, b, gap))
                np.testing.assert_array_equal(np.asarray(chosen)[i], first[i, b])
        return again
    finally:
        np.random.set_state(keep)
