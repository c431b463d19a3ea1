"""GPU: map shifts WITHOUT data movement (circular origin + pending entries that the frame kernels replay) against the oracle's
literal restatement of the reference (roll + pad + z offset, oracle/emap_oracle.py: OracleMap.move_to / move, itself pinned
against the reference's host code in tests/test_oracle_warm_fixtures.py).  Reference: EM/elevation_mapping.py:139-226,
EM/semantic_map.py:127-136; the normal map and traversability_input are NOT shifted there, and are not here."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_close, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu

MOVES = [("move_to", (0.13, -0.3, 0.05)), ("move", (-0.21, 0.09, -0.02)), ("move_to", (0.5, 0.5, 0.0)), ("move", (0.0, 0.0, 0.3)),
         ("move_to", (-0.9, 0.46, 0.11)), ("move", (1.2, -0.04, 0.0)), ("move_to", (0.02, 0.01, 0.2)), ("move", (-0.33, 0.61, -0.07))]


def _move(hip, orc, kind, vec):
    v = np.array(vec, np.float64)
    if kind == "move_to":
        hip.move_to(v, np.eye(3)); orc.move_to(v)
    else:
        hip.move(v); orc.move(v)
    assert np.array_equal(hip.center, orc.center)


def _frame(hip, orc, p, R, t_world, pn=1.0, on=1.0):
    hip.update_map_with_kernel(p, [], R, t_world.copy(), pn, on)
    orc.update_map_with_kernel(p, R, (t_world - orc.center).astype(np.float32), pn, on)


def _check(hip, orc, what):
    assert_planes_close(hip.elevation_map, orc.elevation_map, atol=2e-5, what=what)          # z offsets: float32 add here, float64 in NumPy
    assert_planes_close(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"], what=what + " normals (never shifted)")
    for k in (2, 4, 6):
        assert np.array_equal(hip.elevation_map[k] > 0.5, orc.elevation_map[k] > 0.5), what


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
@pytest.mark.parametrize("cfg_name", ["yaml", "yaml_norays"])
def test_frames_interleaved_with_moves(cfg_name, scatter, weights):
    """move -> frame (the frame replays the pending shift), move -> update_time (written out there), several moves in a row
    (more than the replay list holds), move -> read-back; rays on: the un-shifted normal map feeds the next frame's ray test."""
    C, N = 202, 30000
    cfg = dict(eo.YAML) if cfg_name == "yaml" else dict(eo.YAML, enable_visibility_cleanup=False)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    hip.set_scatter_mode(scatter)
    orc.center = np.zeros(3, np.float32)
    R, t0 = fx.POSES["rotated"]
    tw = lambda: (t0 + hip.center).astype(np.float32)             # sensor rides with the map centre
    _frame(hip, orc, fx.cloud(C, N, 0), R, tw())
    for k in range(7):
        hip.update_time(); orc.update_time()
    _check(hip, orc, "frame 0")
    _move(hip, orc, *MOVES[0])
    _frame(hip, orc, fx.cloud(C, N, 1, dz=-0.02), R, tw())        # pending shift replayed inside the frame
    _check(hip, orc, "move, frame")
    _move(hip, orc, *MOVES[1])
    hip.update_time(); orc.update_time()                            # pending shift written out by the decay pass
    hip.update_variance(); orc.update_variance()
    _check(hip, orc, "move, decay")
    for mv in MOVES[2:8]:                                           # six moves without a frame: the replay list (4) overflows once
        _move(hip, orc, *mv)
    _check(hip, orc, "six moves")                                   # read-back writes the rest out
    for k in range(6):
        hip.update_time(); orc.update_time()
    _frame(hip, orc, fx.cloud(C, N, 2, dz=-0.1), R, tw())
    _move(hip, orc, "move", (0.08, -0.12, 0.01))
    _frame(hip, orc, fx.cloud(C, N, 3, dz=-0.05), R, tw())
    _check(hip, orc, "end")
    assert np.allclose(hip.traversability_input, orc.traversability_input, atol=2e-5)        # (z offsets: see _check)


def test_shift_is_bit_identical_to_a_rolled_copy(weights):
    """the circular origin is invisible: a map that was shifted equals a fresh context loaded with the rolled planes, frame after frame"""
    C, N = 130, 15000
    cfg = dict(eo.YAML)
    a, _ = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    a.update_map_with_kernel(fx.cloud(C, N, 0), [], R, t.copy(), 1.0, 1.0)
    for k in range(7):
        a.update_time()
    a.shift_map_xy(np.array([5, -9])); a.shift_map_z(0.25)
    b, _ = make_pair(cfg, C, "reference_fp16", weights)
    b.elevation_map = a.elevation_map; b.normal_map = a.normal_map
    for f in (1, 2):
        p = fx.cloud(C, N, f, dz=-0.03 * f)
        a.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0); b.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        for k in range(6):
            a.update_time(); b.update_time()
    assert a.elevation_map.tobytes() == b.elevation_map.tobytes()
    assert a.normal_map.tobytes() == b.normal_map.tobytes()
    out_a = np.zeros((C - 2, C - 2), np.float32); out_b = np.zeros((C - 2, C - 2), np.float32)
    for name in ("elevation", "traversability", "upper_bound", "normal_x"):
        a.get_map_with_name_ref(name, out_a); b.get_map_with_name_ref(name, out_b)
        assert out_a.tobytes() == out_b.tobytes(), name


def test_semantic_layers_shift_with_the_map(weights):
    """SemanticMap.shift_map_xy: roll + zero pad of the layers (semantic_map.py:127-136), then fusion on the shifted layers"""
    C, N = 130, 20000
    CH = ["x", "y", "z", "s0", "s1", "c0", "rgb"]
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    orc.center = np.zeros(3, np.float32)
    hip.param.pointcloud_channel_fusions = {"rgb": "color", "c0": "class_average", "default": "average"}
    R, t = fx.POSES["identity"]

    def frame(f):
        p = fx.semantic_cloud(C, N, f)
        hip.input_pointcloud(p, CH, R, (t + hip.center).astype(np.float32), 0.0, 0.0)
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, average=[(3, 0), (4, 1)], class_average=[(5, 2)], color=[(6, 3)], alpha=0.5)

    def check(what):
        sm = hip.semantic_map.semantic_map
        assert np.allclose(sm[:3], orc.semantic_map[:3], atol=1e-6, rtol=1e-5), what
        assert np.array_equal(sm[3].view(np.uint32), orc.semantic_map[3].view(np.uint32)), what
    frame(0)
    _move(hip, orc, "move", (0.16, -0.24, 0.0))
    check("after the move")
    assert (hip.semantic_map.semantic_map[0][:4] == 0).all()       # 0.16 m = 4 rows came in at the top
    frame(1)
    _move(hip, orc, "move_to", (-0.3, 0.5, 0.1))
    frame(2)
    check("frames on shifted layers")
