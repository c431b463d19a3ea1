"""CPU: the exchange plan of the un-shifted normal planes on row strips (emap_api.hip: lag_pieces / normal_exchange, exposed as
emap_normal_lag_plan).  After a row shift the normal of the cell in physical row p lives in physical row (p + lag) mod C of the planes
(the reference does not shift normal_map: EM/elevation_mapping.py:200-214); every rank fetches the rows its cells belong to from whoever
owns them.  Properties: every row of every rank's copy is written exactly once and from the right row; sources are owned rows; both
ends of a pair walk the same pieces in the same order (what lets a grouped send / receive pair up)."""
import ctypes as ct

import numpy as np
import pytest

from elevation_mapping_cupy_amd import _lib
from elevation_mapping_cupy_amd.sharded import ray_balanced_weights, strip_rows


def _plan(C, cuts, lag):
    lib = _lib.load()
    W = len(cuts)
    b = (ct.c_int32 * W)(*[c[0] for c in cuts]); n = (ct.c_int32 * W)(*[c[1] - c[0] for c in cuts])
    out = (ct.c_int32 * (5 * 8 * W))(); k = ct.c_int32(0)
    assert lib.emap_normal_lag_plan(C, W, b, n, lag, out, 8 * W, ct.byref(k)) == 0
    return np.array(out[:5 * k.value], np.int64).reshape(-1, 5)


@pytest.mark.parametrize("C,world,weighted", [(300, 2, False), (300, 8, False), (130, 3, False), (1024, 8, True), (257, 5, False), (4096, 4, False), (202, 1, False)])
def test_every_row_fetched_once_from_where_it_lives(C, world, weighted):
    w = ray_balanced_weights(C, 0.04, 10.0, 7, world) if weighted else None
    cuts = [strip_rows(C, world, r, w) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == C
    rng = np.random.default_rng(C + world)
    lags = [0, 1, -1, 7, -8, 30, -45, C // 2, -(C // 2), C - 1, 1 - C, cuts[0][1], -cuts[0][1], 3 * C + 5] + [int(x) for x in rng.integers(-C, C, 6)]
    for lag in lags:
        P = _plan(C, cuts, lag)
        for q, (b, e) in enumerate(cuts):
            got = np.full(e - b, -1, np.int64)
            for _, r, src, dst, rows in P[P[:, 0] == q]:
                assert rows > 0 and cuts[r][0] <= src and src + rows <= cuts[r][1], "a source piece must lie inside its owner's rows"
                assert np.all(got[dst:dst + rows] == -1), "a row of the copy is written twice"
                got[dst:dst + rows] = src + np.arange(rows)
            assert np.array_equal(got, (b + np.arange(e - b) + lag) % C), "rank %d, lag %d" % (q, lag)
        # both ends of every (sender, receiver) pair see the same sizes in the same order
        for q in range(world):
            for r in range(world):
                if q != r:
                    recv = [int(x[4]) for x in P if x[0] == q and x[1] == r]       # what q posts as receives from r
                    send = [int(x[4]) for x in P if x[1] == r and x[0] == q]       # what r posts as sends to q (same list walk)
                    assert recv == send
        if lag % C == 0:
            assert np.all(P[:, 0] == P[:, 1]), "no shift: everything is local"
