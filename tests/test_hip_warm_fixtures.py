"""GPU: the HIP path (C ABI, both scatter modes) on the race-free warm fixtures of tests/_warm.py, against the committed outputs of
the REFERENCE'S OWN kernels (tests/golden/warm_single.npz) and against the contract oracle: outlier variance inflation, ray
penetration (decrement, cosine test, wall skip) -- reference custom_kernels.py:173-175, 236-258 -- and the host steps of the path
(drift gate, overlap clearance, decay, map shift) against the outputs of the reference's host code (tests/golden/host_steps.npz)."""
import os

import numpy as np
import pytest

import _fixtures as fx
import _warm as W
from _util import make_parameter
from oracle import emap_oracle as eo
from test_oracle_warm_fixtures import GOLD, _close, _host_cases

pytestmark = pytest.mark.gpu


def _hip(cfg, C, scatter, m0, nrm):
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    hip = ElevationMap(make_parameter(cfg, C))
    hip.set_scatter_mode(scatter)
    hip.elevation_map = m0; hip.normal_map = nrm
    return hip


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
@pytest.mark.parametrize("name", list(W.SETS))
def test_single_point_warm_frames_vs_reference_golden(name, scatter):
    g = np.load(os.path.join(GOLD, "warm_single.npz"))
    cfg = getattr(eo, W.SETS[name]); C = 202
    m0, nrm = fx.warm_map(C, 1, cfg["initial_variance"])
    hip = _hip(cfg, C, scatter, m0, nrm)
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C)); om.elevation_map[...] = m0; om.normal_map[...] = nrm
    k = [0]

    def frame(p, R, t):
        W.hip_frame(hip, p, R, t); W.oracle_frame(om, p, R, t)
        k[0] += 1
        if k[0] % 50 == 0:
            _close(hip.elevation_map, om.elevation_map, "%s/%s frame %d vs oracle" % (name, scatter, k[0]))

    def tick():
        hip.update_time(); om.update_time()
    W.run_single(cfg, C, 1, frame, None, tick)
    want = W.apply_sparse(W.base_single(m0, cfg), g[name + "_idx"], g[name + "_val"])
    _close(hip.elevation_map, want, "%s/%s final vs reference golden" % (name, scatter))
    assert g[name + "_hits_outliers"][0] >= 1000 and g[name + "_hits_outliers"][1] >= 20


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_wall_skip_fixture_vs_reference_golden(scatter):
    g = np.load(os.path.join(GOLD, "warm_single.npz"))
    cfg = dict(eo.DEFAULTS, **W.WALL_CFG); C = 202
    m0, nrm = fx.warm_map(C, 2, cfg["initial_variance"])
    hip = _hip(cfg, C, scatter, m0, nrm)
    R, t = fx.POSES["identity"]
    cur, curn = m0.copy(), nrm.copy()
    va = []
    for ix, iy, cell, d3, pts, skipped in W.wall_sequence(C):
        # inject the cell through the plane interface: read-modify-write of the host mirror
        cur = hip.elevation_map; cur[:, ix, iy] = cell; hip.elevation_map = cur
        curn[:, ix, iy] = d3; hip.normal_map = curn
        W.hip_frame(hip, pts, R, t)
        v = hip.get_layer_raw(2)[ix, iy]
        assert (v == 1.0) == skipped
        va.append(v)
    want = W.apply_sparse(W.base_after_reset(m0, np.float32(cfg["initial_variance"])), g["wall202_idx"], g["wall202_val"])
    _close(hip.elevation_map, want, "wall fixture (%s) vs reference golden" % scatter)
    assert np.allclose(np.array(va, np.float32), g["wall202_valid_after"], atol=1e-6)


@pytest.mark.parametrize("cname", ["YAML", "DEFAULTS"])
def test_host_steps_vs_reference_host_code_golden(cname):
    """drift gate on the device (k_gate), overlap clearance, variance / time decay == outputs of the reference's host code"""
    g = np.load(os.path.join(GOLD, "host_steps.npz"))
    cfg = dict(getattr(eo, cname))
    gate_cases, _ = _host_cases()
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    for row, (err, cnt, pn, on) in zip(g[cname + "_gate"], gate_cases):
        hip = ElevationMap(make_parameter(cfg, 34))
        hip.set_layer_raw(0, fx.stencil_inputs(34, 3)[0])
        hip.stage("gate", position_noise=pn, orientation_noise=on, err_sum=float(np.float32(err)), err_cnt=int(cnt))
        hip.stage("average")                                   # materialises the shift (elevation_mapping.py:357)
        st = hip.stats()
        fired = row[1] != 0.25
        assert bool(st.gate_fired) == fired
        if fired:
            assert abs(st.mean_error - row[0]) <= 1e-7 and abs(st.additive_mean_error - (row[1] - 0.25)) <= 1e-6
        # cell (5, 7) is unknown on this map => average_map resets it; compare the shift itself
        shift_want = np.float32(row[2]) - fx.stencil_inputs(34, 3)[0][5, 7]
        assert abs(st.shift - shift_want) <= 2e-7, (err, cnt, pn, on)
        hip.close()
    C = 130
    hip = ElevationMap(make_parameter(cfg, C))
    m0, _ = fx.warm_map(C, 3, cfg["initial_variance"]); hip.elevation_map = m0
    hip.stage("overlap", t=float(np.float32(2.6)))
    want = W.apply_sparse(m0, g[cname + "_overlap_idx"], g[cname + "_overlap_val"])
    got = hip.elevation_map
    assert all(np.array_equal(got[q], want[q]) for q in range(7)), "overlap clearance must be exact"
    hip.update_variance(); hip.update_time()
    assert np.array_equal(hip.elevation_map[[1, 4]], g[cname + "_decay_var_time"])


def test_map_shift_vs_reference_host_code_golden():
    """move_to / move sequence == the reference's host code (roll + pad, z shift), centre included"""
    g = np.load(os.path.join(GOLD, "host_steps.npz"))
    _, moves = _host_cases()
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    cfg = dict(eo.YAML); C = 34
    hip = ElevationMap(make_parameter(cfg, C))
    m0, _ = fx.warm_map(C, 4, cfg["initial_variance"]); hip.elevation_map = m0
    for (kind, vec), c_want in zip(moves, g["shift_centers"]):
        if kind == "move_to":
            hip.move_to(np.array(vec, np.float64), np.eye(3))
        else:
            hip.move(np.array(vec, np.float64))
        assert np.allclose(hip.center, c_want, atol=1e-6)
    got, want = hip.elevation_map, g["shift_map"]
    for q in range(7):
        assert np.allclose(got[q], want[q], atol=1e-6, rtol=1e-6), q     # z shift: float32 add here, one float64 rounding in NumPy
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[4], want[4]) and np.array_equal(got[6], want[6])
