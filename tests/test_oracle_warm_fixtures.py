"""CPU: the contract oracle on the RACE-FREE warm fixtures of tests/_warm.py, pinned two ways:
  * live against the reference's own kernel source compiled for the host (oracle/_ref), frame by frame, all 7 planes;
  * against the committed outputs of that same run (tests/golden/warm_single.npz, tests/golden/make_golden.py::warm_single),
    so the pin also holds where neither /root/reference nor oracle/_ref exist.
These fixtures reach what a fresh map cannot: the outlier variance inflation (reference custom_kernels.py:173-175), the ray
penetration branch (:236-258) with its decrement, cosine test and wall skip.  Host steps of the path (drift gate, overlap
clearance, variance / time decay, map shift) are pinned the same way against the reference's HOST code executed with NumPy as
cupy (oracle/ref_host.py -> tests/golden/host_steps.npz)."""
import os

import numpy as np
import pytest

import _fixtures as fx
import _warm as W
from oracle import build_ref, emap_oracle as eo, ref_host, ref_kernels

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLANES = ["elevation", "variance", "is_valid", "traversability", "time", "upper_bound", "is_upper_bound"]


def _ref(name):
    p = build_ref.PREBUILD[name]
    if not ref_kernels.available(p):
        pytest.skip("compiled reference object not available (no /root/reference and not prebuilt)")
    return ref_kernels.RefKernels(p)


def _close(a, b, what):
    for q in range(7):
        bad = ~np.isclose(a[q], b[q], atol=1e-5, rtol=1e-5)
        assert not bad.any(), "%s: plane %s differs on %d cells (max |d| %g)" % (what, PLANES[q], int(bad.sum()), np.abs(a[q] - b[q]).max())
    for q in (2, 4, 6):                       # flags and the time plane are exact wherever no ray decrement accumulated
        same = a[q] == b[q]
        assert same.all() or q == 2, "%s: plane %s not exact" % (what, PLANES[q])


def _oracle_single(name):
    cfg = getattr(eo, W.SETS[name])
    C = 202
    m0, nrm = fx.warm_map(C, 1, cfg["initial_variance"])
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C))
    om.elevation_map[...] = m0; om.normal_map[...] = nrm
    return cfg, C, m0, nrm, om


@pytest.mark.parametrize("name", list(W.SETS))
def test_single_point_warm_frames_vs_reference_source(name):
    """400 one-point frames: oracle == compiled reference after EVERY frame on all 7 planes; the branches were reached."""
    rk = _ref(name)
    cfg, C, m0, nrm, om = _oracle_single(name)
    m = m0.copy()
    tot = [0, 0, 0]

    def frame(p, R, t):
        W.ref_frame(rk, m, nrm, p, R, t)
        h, o = W.oracle_frame(om, p, R, t)
        tot[0] += h; tot[1] += o; tot[2] += 1
        _close(om.elevation_map, m, "%s frame %d" % (name, tot[2] - 1))

    def tick():
        m[4] += np.float32(cfg["time_interval"]); om.update_time()
    W.run_single(cfg, C, 1, frame, None, tick)
    assert tot[2] == W.K_FRAMES and tot[0] >= 1000 and tot[1] >= 20, tot


@pytest.mark.parametrize("name", list(W.SETS))
def test_single_point_warm_frames_vs_golden(name):
    g = np.load(os.path.join(GOLD, "warm_single.npz"))
    cfg, C, m0, nrm, om = _oracle_single(name)
    tot = [0, 0]

    def frame(p, R, t):
        h, o = W.oracle_frame(om, p, R, t); tot[0] += h; tot[1] += o
    W.run_single(cfg, C, 1, frame, None, om.update_time)
    want = W.apply_sparse(W.base_single(m0, cfg), g[name + "_idx"], g[name + "_val"])
    _close(om.elevation_map, want, name + " final")
    assert tot == list(g[name + "_hits_outliers"]) and tot[0] >= 1000 and tot[1] >= 20


def _wall_run(on_frame):
    cfg = dict(eo.DEFAULTS, **W.WALL_CFG)
    C = 202
    m0, nrm = fx.warm_map(C, 2, cfg["initial_variance"])
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C))
    om.elevation_map[...] = m0; om.normal_map[...] = nrm
    R, t = fx.POSES["identity"]
    valid_after = []
    for k, (ix, iy, cell, d3, pts, skipped) in enumerate(W.wall_sequence(C)):
        om.elevation_map[:, ix, iy] = cell; om.normal_map[:, ix, iy] = d3
        on_frame(k, ix, iy, cell, d3, pts, R, t)
        W.oracle_frame(om, pts, R, t)
        assert om.last["n_inl"][ix, iy] == (2 if skipped else 0) and om.last["cnt"][ix, iy] == 0   # inliers counted, nothing fused
        assert (om.elevation_map[2, ix, iy] == 1.0) == skipped, "frame %d: wall skip %s expected" % (k, skipped)
        valid_after.append(om.elevation_map[2, ix, iy])
    return cfg, m0, om, np.array(valid_after, np.float32)


def test_wall_skip_fixture_vs_reference_source():
    """`newmap[3] > wall_num_thresh && time < 1.0` (reference custom_kernels.py:246-247): 60 skipped / 60 penetrated cells"""
    rk = _ref("wall202")
    state = {}

    def on_frame(k, ix, iy, cell, d3, pts, R, t):
        if not state:
            state["m"], state["n"] = fx.warm_map(202, 2, eo.DEFAULTS["initial_variance"])
        state["m"][:, ix, iy] = cell; state["n"][:, ix, iy] = d3
        W.ref_frame(rk, state["m"], state["n"], pts, R, t)
    cfg, m0, om, va = _wall_run(on_frame)
    _close(om.elevation_map, state["m"], "wall fixture final")
    assert int((va == 1).sum()) == 60 and int((va < 1).sum()) == 60


def test_wall_skip_fixture_vs_golden():
    g = np.load(os.path.join(GOLD, "warm_single.npz"))
    cfg, m0, om, va = _wall_run(lambda *a: None)
    want = W.apply_sparse(W.base_after_reset(m0, np.float32(cfg["initial_variance"])), g["wall202_idx"], g["wall202_val"])
    _close(om.elevation_map, want, "wall fixture final")
    assert np.allclose(va, g["wall202_valid_after"], atol=1e-6)


# ---- host steps --------------------------------------------------------------------------------------------------------------
def _gate_oracle(cfg, err, cnt, pn, on):
    om = eo.OracleMap(eo.make_params(cfg, cell_n=34))
    om.elevation_map[0] = fx.stencil_inputs(34, 3)[0]
    om.additive_mean_error = np.float32(0.25)
    om.last.update(err_sum=float(np.float32(err)), err_cnt=int(cnt))
    om.gate(pn, on)
    return [float(om.mean_error), float(om.additive_mean_error), float(om.elevation_map[0, 5, 7])]


def _host_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mk_golden", os.path.join(GOLD, "make_golden.py"))
    src = open(spec.origin).read()
    ns = {}
    for name in ("GATE_CASES", "MOVE_SEQUENCE"):            # the two literal tables of the generator (no import: it needs /root/reference)
        start = src.index(name + " = ")
        end = src.index("]\n", start) + 1
        exec(src[start:end], ns)
    return ns["GATE_CASES"], ns["MOVE_SEQUENCE"]


@pytest.mark.parametrize("cname", ["YAML", "DEFAULTS"])
def test_host_steps_vs_golden_of_reference_host_code(cname):
    g = np.load(os.path.join(GOLD, "host_steps.npz"))
    cfg = dict(getattr(eo, cname))
    gate_cases, _ = _host_cases()
    for row, (err, cnt, pn, on) in zip(g[cname + "_gate"], gate_cases):
        assert np.allclose(_gate_oracle(cfg, err, cnt, pn, on), row, atol=1e-7, rtol=1e-7), (err, cnt, pn, on)
    C = 130
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C))
    m0, _ = fx.warm_map(C, 3, cfg["initial_variance"]); om.elevation_map[...] = m0
    om.overlap_clear(float(np.float32(2.6)))
    want = W.apply_sparse(m0, g[cname + "_overlap_idx"], g[cname + "_overlap_val"])
    assert all(np.array_equal(om.elevation_map[q], want[q]) for q in range(7)), "overlap clearance must be exact"
    assert g[cname + "_overlap_idx"].size > 1000
    om.update_variance(); om.update_time()
    assert np.array_equal(om.elevation_map[[1, 4]], g[cname + "_decay_var_time"])


def test_map_shift_vs_golden_of_reference_host_code():
    g = np.load(os.path.join(GOLD, "host_steps.npz"))
    _, moves = _host_cases()
    cfg = dict(eo.YAML); C = 34
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C))
    m0, _ = fx.warm_map(C, 4, cfg["initial_variance"]); om.elevation_map[...] = m0
    for (kind, vec), c_want in zip(moves, g["shift_centers"]):
        getattr(om, kind)(np.array(vec, np.float64))
        assert np.array_equal(om.center, c_want)
    assert np.array_equal(om.elevation_map, g["shift_map"])


def test_host_steps_live_against_reference_host_code():
    """same comparisons with the reference file executed here (only where /root/reference exists)"""
    if not ref_host.available():
        pytest.skip("/root/reference not present")
    H = ref_host.load()
    gate_cases, moves = _host_cases()
    for cname in ("YAML", "DEFAULTS"):
        cfg = dict(getattr(eo, cname))
        for err, cnt, pn, on in gate_cases:
            h = H(cfg, 34); h.elevation_map[0] = fx.stencil_inputs(34, 3)[0]; h.additive_mean_error = np.float32(0.25)
            h.drift_gate(np.array([err], np.float32), np.array([cnt], np.float32), pn, on)
            got = _gate_oracle(cfg, err, cnt, pn, on)
            want = [float(np.asarray(h.mean_error).ravel()[0]), float(np.asarray(h.additive_mean_error).ravel()[0]), float(h.elevation_map[0, 5, 7])]
            assert np.allclose(got, want, atol=1e-7, rtol=1e-7)
    cfg = dict(eo.YAML); C = 34
    h = H(cfg, C); om = eo.OracleMap(eo.make_params(cfg, cell_n=C))
    m0, _ = fx.warm_map(C, 4, cfg["initial_variance"]); h.elevation_map[...] = m0; om.elevation_map[...] = m0
    for kind, vec in moves:
        getattr(h, kind)(*((np.array(vec, np.float64), np.eye(3)) if kind == "move_to" else (np.array(vec, np.float64),)))
        getattr(om, kind)(np.array(vec, np.float64))
        assert np.array_equal(np.asarray(h.center, np.float32), om.center) and np.array_equal(h.elevation_map, om.elevation_map)


@pytest.mark.parametrize("name,seed", [("yaml202", 2), ("default202", 3)])
def test_fuzz_single_point_warm_frames_under_random_poses_vs_reference_source(name, seed):
    """the race-free one-point frames once more -- another injected warm map, 300 frames whose sensor pose changes every ten frames
    (any yaw, +-60 degrees of roll / pitch, up to 1.5 m off the map centre, sensor height 0.5 ... 2 m): oracle == compiled reference
    after EVERY frame on all 7 planes, with the ray-penetration and outlier branches reached."""
    rk = _ref(name)
    cfg = getattr(eo, W.SETS[name])
    C = 202
    m0, nrm = fx.warm_map(C, seed, cfg["initial_variance"])
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C))
    om.elevation_map[...] = m0; om.normal_map[...] = nrm
    m = m0.copy()
    rng = np.random.default_rng(50 + seed)
    pts = fx.cloud(C, 300, 7100 + seed)
    hits = outl = 0
    R = t = None
    for k in range(300):
        if k % 10 == 0:
            a = rng.uniform(-np.pi, np.pi, 3) * np.array([0.33, 0.33, 1.0])
            R = np.ascontiguousarray(fx.rot(a[0], a[1], a[2]), np.float32)
            t = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(0.5, 2.0)], np.float32)
            m[4] += np.float32(cfg["time_interval"]); om.update_time()
        p = pts[k:k + 1].copy()
        W.ref_frame(rk, m, nrm, p, R, t)
        h, o = W.oracle_frame(om, p, R, t)
        hits += h; outl += o
        _close(om.elevation_map, m, "%s frame %d" % (name, k))
    assert hits >= 300 and outl >= 5, (hits, outl)
