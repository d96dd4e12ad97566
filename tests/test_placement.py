"""The XCD placement of single-round launches (csrc/l2a_mfma.h geometry, csrc/l2a_api.hip launch_rollout): a Python mirror of
the index arithmetic - every logical workgroup must be run by exactly one hardware workgroup, the spare ones must all
return, and a unit's workgroups must sit on that unit's XCDs only (hardware workgroup i runs on XCD i % 8)."""

import pytest


def _host(units, w, cus=256):
    """launch_rollout: (pl_units, pl_f, pl_r, pl_w, grid) or None when the contiguous remap stays."""
    if not 2 <= units <= 8:
        return None
    f = 8 // units
    slots = (w + f - 1) // f
    if 8 * slots > cus:
        return None
    return units, f, 8 - units * f, w, 8 * slots


def _kernel(hw, pl):
    """l2a_rollout_mfma_k: hardware workgroup -> (unit, member) or None (returns at once)."""
    units, f, r, w, _ = pl
    xcd, idx = hw & 7, hw >> 3
    wide = r * (f + 1)
    if xcd < wide:
        u, k, xu = xcd // (f + 1), xcd % (f + 1), f + 1
    else:
        y = xcd - wide
        u, k, xu = r + y // f, y % f, f
    su = (w + xu - 1) // xu
    j = k * su + idx
    if idx >= su or j >= w:
        return None
    return u, j, xcd


@pytest.mark.parametrize("units", range(2, 9))
def test_every_member_of_every_unit_runs_exactly_once(units):
    for w in list(range(1, 70)) + [96, 125, 127, 128]:
        pl = _host(units, w)
        if pl is None:
            continue
        seen, xcds = {}, {}
        for hw in range(pl[4]):
            hit = _kernel(hw, pl)
            if hit is None:
                continue
            u, j, xcd = hit
            assert 0 <= u < units and 0 <= j < w
            assert (u, j) not in seen, "unit %d member %d twice (units %d, w %d)" % (u, j, units, w)
            seen[(u, j)] = hw
            xcds.setdefault(u, set()).add(xcd)
        assert len(seen) == units * w, "units %d w %d: %d of %d placed" % (units, w, len(seen), units * w)
        owned = [x for s in xcds.values() for x in s]
        assert len(owned) == len(set(owned)), "two units share an XCD (units %d, w %d)" % (units, w)


def test_config_shapes():
    # config 2: two ensemble groups x 125 tiles -> 256 hardware workgroups, A on XCDs 0-3, B on 4-7
    pl = _host(2, 125)
    assert pl == (2, 4, 0, 125, 256)
    groups = {0: set(), 1: set()}
    for hw in range(256):
        hit = _kernel(hw, pl)
        if hit:
            groups[hit[0]].add(hit[2])
    assert groups == {0: {0, 1, 2, 3}, 1: {4, 5, 6, 7}}
    # config 3b: five environments x 32 tiles -> three of them on two XCDs, two on one
    pl = _host(5, 32)
    assert pl == (5, 1, 3, 32, 256)
    # too wide for one round: the contiguous remap stays
    assert _host(2, 129) is None and _host(9, 4) is None
