"""CPU: the C oracle against the reference's own kernels compiled for the host (oracle/_ref, built from
/root/reference by oracle/build_ref.py; prebuilt objects travel with the repo snapshot).  Order-independent
outputs are compared on ARBITRARY (warm, racy) inputs; race-exposed planes only on race-free fixtures
(DESIGN.md "Determinism contract")."""
import numpy as np
import pytest

import _fixtures as fx
from oracle import build_ref, emap_oracle as eo, ref_kernels


def _ref(name):
    p = build_ref.PREBUILD[name]
    if not ref_kernels.available(p):
        pytest.skip("compiled reference object not available (no /root/reference and not prebuilt)")
    return ref_kernels.RefKernels(p)


def _warm(cfg, C, N, weights):
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    R, t = fx.POSES["rotated"]
    om.update_map_with_kernel(fx.cloud(C, N, 0), R, t)
    for _ in range(10):
        om.update_time()
    om.update_variance()
    return om, R.ravel().copy(), t.copy()


@pytest.mark.parametrize("name,cfg", [("yaml202", eo.YAML), ("default202", eo.DEFAULTS)])
def test_count_outputs_on_warm_map(name, cfg, weights):
    """error_counting_kernel only reads the map => exact comparison on an arbitrary warm map."""
    rk = _ref(name)
    C, N = 202, 50000
    om, R, t = _warm(cfg, C, N, weights)
    p = fx.cloud(C, N, 1, dz=-0.01)
    nm = np.zeros((7, C, C), np.float32); err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
    rk.error_counting(om.elevation_map.copy(), p.copy(), R, t, nm, err, cnt)
    n_pts, n_inl, es, ec = om.count(p, R, t)
    assert np.array_equal(n_pts, nm[4].astype(np.uint32)) and np.array_equal(n_inl, nm[3].astype(np.uint32))
    assert ec == int(cnt[0])
    assert abs(es - float(err[0])) <= 1e-4 * max(1.0, ec) * 1e-1 + 1e-5      # reference accumulates in float32, order dependent


@pytest.mark.parametrize("name,cfg", [("yaml202", eo.YAML), ("default202", eo.DEFAULTS)])
def test_count_outputs_on_the_terrain_scene(name, cfg, weights):
    """the same comparison on a spatially coherent, scan-ordered cloud (tests/_fixtures.py: terrain_cloud): hundreds of points per cell
    next to the sensor, most of the map never seen -- the inputs of tests/test_hip_terrain.py, so the oracle those GPU tests compare
    with is tied to the reference's kernel on this kind of cloud too."""
    rk = _ref(name)
    C = 202
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    R, t = np.eye(3, dtype=np.float32), np.array([0, 0, 1], np.float32)
    om.update_map_with_kernel(fx.terrain_cloud(C, 300, 160, 0), R, t)
    for _ in range(10):
        om.update_time()
    om.update_variance()
    p = fx.terrain_cloud(C, 300, 160, 1, shift=1.5)
    Rf = R.ravel().copy()
    nm = np.zeros((7, C, C), np.float32); err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
    rk.error_counting(om.elevation_map.copy(), p.copy(), Rf, t.copy(), nm, err, cnt)
    n_pts, n_inl, es, ec = om.count(p, R, t)
    assert int(n_pts.max()) > 100 and int((n_pts > 0).sum()) < C * C // 2          # piled up, and most cells untouched
    assert np.array_equal(n_pts, nm[4].astype(np.uint32)) and np.array_equal(n_inl, nm[3].astype(np.uint32))
    assert ec == int(cnt[0]) and ec > 100
    assert abs(es - float(err[0])) <= 1e-4 * max(1.0, ec) * 1e-1 + 1e-5      # reference accumulates in float32, order dependent


def test_fuse_sums_without_outliers_rays_off(weights):
    """F2 fixture: rays disabled, fresh cells only => sequential == contract for every plane."""
    rk = _ref("yaml202_norays")
    C, N = 202, 50000
    cfg = dict(eo.YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    R, t = fx.POSES["rotated"]; Rf = R.ravel().copy()
    p = fx.cloud(C, N, 4)
    m = om.elevation_map.copy(); nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
    err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32); pr = p.copy()
    rk.error_counting(m, pr, Rf, t, nm, err, cnt); rk.add_points(Rf, t, nrm, pr, m, nm); rk.average_map(nm, m)
    om.count(p, R, t); om.gate(0, 0); om.fuse(p, R, t); om.commit(); om.average()
    for pl in (2, 3, 4, 5, 6):
        assert np.array_equal(om.elevation_map[pl], m[pl]), pl
    assert np.allclose(om.elevation_map[0], m[0], atol=1e-5, rtol=1e-5)
    assert np.allclose(om.elevation_map[1], m[1], atol=1e-5, rtol=1e-5)


def test_warm_frame_race_exposed_planes_differ_only_where_the_reference_races(weights):
    """On a warm map the sequential run and the snapshot contract legitimately differ (SURVEY §8a'): every differing
    cell must be in one of the reference's race classes (custom_kernels.py:170-192 vs :213-256): r1/r2 multi-point
    cell or a cell with an outlier; r3 a stale valid cell that is fused this frame (in the sequential run earlier rays
    inflate its variance before its point arrives); or a cell the visibility pass writes."""
    rk = _ref("yaml202")
    C, N = 202, 50000
    om, R, t = _warm(eo.YAML, C, N, weights)
    p = fx.cloud(C, N, 1, dz=-0.02)
    pre = om.elevation_map.copy()
    m = pre.copy(); nm = np.zeros((7, C, C), np.float32); nrm = om.normal_map.copy()
    err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32); pr = p.copy()
    rk.error_counting(m, pr, R, t, nm, err, cnt); rk.add_points(R, t, nrm, pr, m, nm); rk.average_map(nm, m)
    om.count(p, R, t); om.gate(0, 0); om.fuse(p, R, t); om.commit(); om.rays(p, R, t)
    racy = (om.last["n_pts"] > 1) | (om.last["n_out"] > 0) | (om.last["ray_hits"] > 0) | np.isfinite(om.last["ray_upper"])
    racy |= (pre[2] > 0.5) & (pre[4] >= 0.5) & (om.last["n_pts"] > 0)          # r3
    assert racy.mean() < 0.9
    om.average()
    for pl in range(7):
        diff = ~np.isclose(om.elevation_map[pl], m[pl], atol=1e-5, rtol=1e-5)
        assert not (diff & ~racy).any(), "plane %d differs on %d race-free cells" % (pl, int((diff & ~racy).sum()))


@pytest.mark.parametrize("d", [1, 2])
def test_min_filter_single_sweep_on_isolated_holes(d):
    """The MinFilter kernel reads the buffers it writes; with isolated holes (no other hole inside a hole's window)
    the sequential reference and the Jacobi contract coincide exactly, incl. the flat-index wrap at the edges."""
    rk = _ref("default34")
    C = 34
    rng = np.random.default_rng(d)
    e0 = rng.uniform(-1, 1, (C, C)).astype(np.float32)
    v = np.ones((C, C), np.float32)
    v[3::7, 2::6] = 0; v[10, 0] = 0; v[20, C - 1] = 0; v[0, 15] = 0
    newmap, newmask = e0.copy(), v.copy()
    rk.min_filter_sweep(e0, v, newmap, newmask, d)
    want = np.where(newmask > 0.5, newmap, np.nan)
    got, sweeps = eo.min_filter(C, d, 1, e0, v)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


@pytest.mark.parametrize("dist", [False, True])
def test_image_correspondence_and_fusions(dist, weights):
    """camera path: projection + radtan distortion + Bresenham occlusion walk and the two samplers are per-cell and race
    free => exact comparison with the reference kernels on a warm map."""
    rk = _ref("image98")
    C = 98
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    R0, t0 = fx.POSES["identity"]
    p = fx.cloud(C, 20000, 0); p[:, 2] += 0.3 * np.sin(p[:, 0] * 2.0)           # relief => occlusions
    om.update_map_with_kernel(p, R0, t0)
    K, D, R, t, H, W = fx.camera_case(C, 1, dist)
    center = np.array([0.1, -0.2, 0.05], np.float32)
    Pm, x1, y1, z1 = fx.camera_inputs(center, C, 0.04, K, R, t)
    uv_r = np.zeros((2, C, C), np.float32); va_r = np.zeros((C, C), np.bool_)
    rk.image_correspondence(om.elevation_map.copy(), x1, y1, z1, Pm.ravel().copy(), K.ravel().copy(), D.copy(), H, W, center, uv_r, va_r)
    uv, va = eo.image_correspondence(om.P, om.elevation_map, x1, y1, z1, Pm.ravel(), K.ravel(), D, H, W, center)
    assert va_r.sum() > 50 and (uv_r[0] != 0).sum() > va_r.sum()                # visible cells exist, occluded ones too
    assert np.array_equal(va.astype(bool), va_r) and np.array_equal(uv, uv_r)
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    rgb = rng.integers(0, 256, (3, H, W)).astype(np.float32)
    sem = rng.uniform(0, 1, (2, C, C)).astype(np.float32)
    new_r = np.zeros_like(sem)
    rk.image_fuse("exponential", sem.copy(), 0, img[1].copy(), uv_r, va_r, H, W, new_r)
    rk.image_fuse("color", sem.copy(), 1, rgb.copy(), uv_r, va_r, H, W, new_r)
    mine = sem.copy()
    eo.image_fuse(om.P, "exponential", mine[0], img[1], uv, va, H, W, 0.7)
    eo.image_fuse(om.P, "color", mine[1], rgb, uv, va, H, W)
    assert np.array_equal(mine[0], new_r[0]) and np.array_equal(mine[1].view(np.uint32), new_r[1].view(np.uint32))
    # average_correspondences_to_map_kernel (custom_image_kernels.py:160-192; defined by the reference, launched by none of its fusions)
    if hasattr(rk.lib, "ref_image_average"):
        new_a = np.zeros_like(sem)
        rk.image_fuse("average", sem.copy(), 0, img[2].copy(), uv_r, va_r, H, W, new_a)
        mine_a = sem[0].copy()
        eo.image_fuse(om.P, "average", mine_a, img[2], uv, va, H, W)
        assert np.array_equal(mine_a, new_a[0]) and (mine_a != sem[0]).sum() == va_r.sum()
    else:
        pytest.fail("oracle/_ref was built before image_average was added: rebuild (python -m oracle.build_ref)")


@pytest.mark.parametrize("center", [(0.0, 0.0), (0.52, -0.28)])
def test_polygon_mask_restatement_vs_reference_kernel(center):
    """polygon_mask_kernel (custom_kernels.py:509-651): convex, concave, sub-cell and map-filling polygons, bit exact; vertices
    and edges lie on the mask (the kernel's early `return` from its edge loop)."""
    params = build_ref.PREBUILD["polygon130"]
    if not ref_kernels.available(params):
        pytest.skip("compiled reference not built")
    rk = ref_kernels.RefKernels(params, build=False)
    C = 130
    P = eo.make_params(eo.DEFAULTS, cell_n=C)
    polys = [np.array([[-0.8, -0.5], [0.9, -0.7], [1.1, 0.6], [-0.2, 1.2]], np.float32),
             np.array([[-1.5, -1.5], [1.5, -1.5], [1.5, 1.5], [0.0, 0.2], [-1.5, 1.5]], np.float32),
             np.array([[0.3, 0.3], [0.31, 0.3], [0.3, 0.31]], np.float32),
             np.array([[-2.5, -2.5], [2.5, -2.5], [2.5, 2.5], [-2.5, 2.5]], np.float32)]
    for poly in polys:
        poly = (poly + np.array(center, np.float32)).astype(np.float32)
        want = np.full((C, C), -1, np.float32)
        bbox = np.concatenate([poly.min(axis=0), poly.max(axis=0)]).astype(np.float32)
        rk.polygon_mask(poly, center[0], center[1], bbox, want)
        got = eo.polygon_mask(P, poly, center[0], center[1])
        assert np.array_equal(got, want)


@pytest.mark.parametrize("d", [1, 2])
def test_max_filter_sweeps_vs_reference_kernel(d):
    """MaxFilter (plugins/max_filter.py): the reference passes COPIES as inputs, so its sweeps are out of place and the compiled
    kernel driven like MaxFilter.__call__ (:95-112) must equal the restatement exactly -- large holes, several sweeps, early stop."""
    rk = _ref("maxfilter34")
    C = 34
    rng = np.random.default_rng(10 + d)
    e0 = rng.uniform(-1, 1, (C, C)).astype(np.float32)
    v = (rng.uniform(0, 1, (C, C)) > 0.55).astype(np.float32)
    v[5:14, 6:17] = 0                                           # a hole that needs several sweeps
    for iters in (1, 3, 30):
        cur, mask = e0.copy(), v.copy()
        ran = 0
        for _ in range(iters):
            rk.max_filter_sweep(cur.copy(), mask.copy(), cur, mask, d)
            ran += 1
            if (mask > 0.5).all():
                break
        want = np.where(mask > 0.5, cur, np.nan)
        got, sweeps = eo.max_filter(C, d, iters, e0, v)
        assert sweeps == ran
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


def _random_pose(rng):
    a = rng.uniform(-np.pi, np.pi, 3) * np.array([0.35, 0.35, 1.0])
    R = fx.rot(a[0], a[1], a[2])
    t = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 2.5), rng.uniform(0.3, 2.0)], np.float32)
    return np.ascontiguousarray(R, np.float32), t


@pytest.mark.parametrize("name,cfg", [("yaml202", eo.YAML), ("default202", eo.DEFAULTS)])
def test_fuzz_count_outputs_and_point_index_vs_reference_source(name, cfg, weights):
    """Seeded sweep over what the fixed fixtures do not vary: random sensor poses (any yaw, +-60 degrees of roll / pitch, sensor up to
    2.5 m off the map centre), clouds with points far outside the map (clamped to its border by the half-precision index arithmetic),
    points at the sensor (min_valid_distance), large magnitudes (beyond the float16 range: +-inf after the parameter rounding) and
    exact cell-boundary coordinates.  Order-independent outputs only: per-cell point / inlier counts and the drift count of
    error_counting_kernel (custom_kernels.py:280-345) on a warm map, and the (cell index, is_valid, is_inside) triple add_points_kernel
    leaves in the point's first three columns (:260-262) -- all exact."""
    rk = _ref(name)
    C, N = 202, 20000
    om, _, _ = _warm(cfg, C, 50000, weights)
    res = float(eo.DEFAULTS["resolution"] if "resolution" not in cfg else cfg["resolution"])
    rng = np.random.default_rng(2025)
    for case in range(10):
        R, t = _random_pose(rng)
        Rf = R.ravel().copy()
        p = fx.cloud(C, N, 100 + case, dz=float(rng.uniform(-0.3, 0.3)))
        k = rng.integers(0, N, 600)
        p[k[:150], :2] *= np.float32(3.0)                                            # far outside the map
        p[k[150:250]] = rng.normal(0, 0.05, (100, 3)).astype(np.float32)             # at the sensor
        p[k[250:300], int(rng.integers(0, 3))] = np.float32(7.0e4) * rng.choice([-1, 1])      # beyond the largest half (65504)
        p[k[300:450], 0] = (np.round(p[k[300:450], 0] / res) * res).astype(np.float32)        # on cell boundaries
        p[k[450:600], 1] = (np.round(p[k[450:600], 1] / res) * res + np.float32(res / 2)).astype(np.float32)
        nm = np.zeros((7, C, C), np.float32); err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
        rk.error_counting(om.elevation_map.copy(), p.copy(), Rf, t.copy(), nm, err, cnt)
        n_pts, n_inl, es, ec = om.count(p, R, t)
        assert np.array_equal(n_pts, nm[4].astype(np.uint32)), "case %d: points per cell" % case
        assert np.array_equal(n_inl, nm[3].astype(np.uint32)), "case %d: drift inliers per cell" % case
        assert ec == int(cnt[0]), "case %d" % case
        assert abs(es - float(err[0])) <= 1e-5 * max(1.0, ec) + 1e-5
        # the index tail of add_points_kernel on a FRESH map (so that nothing else of the kernel depends on the map's state)
        pr = p.copy()
        m = np.zeros((7, C, C), np.float32); m[1] = cfg.get("initial_variance", eo.DEFAULTS["initial_variance"]); m[3] = 1
        rk.add_points(Rf, t.copy(), np.zeros((3, C, C), np.float32), pr, m, np.zeros((7, C, C), np.float32))
        idx, valid, inside = om.point_index(p, R, t)
        # (points beyond the half range become +-inf / NaN: their int cast is undefined in C++ -- the host-compiled reference gets
        # INT_MIN from x86, the GPU saturates (NaN -> 0) and so do oracle and kernels.  Their is_valid flag is still compared -- a point
        # infinitely far BELOW the sensor passes every height gate on both sides -- and none of them may land inside the map: the
        # counts above are exact with them in the cloud.  Only their meaningless index / border flag is left out.)
        fin = np.abs(p).max(axis=1) < 6.0e4
        assert (~fin).sum() >= 40 and not inside[~fin].any()
        assert np.array_equal(idx[fin], pr[fin, 0].astype(np.int32)), "case %d: cell index" % case
        assert np.array_equal(valid.astype(bool), pr[:, 1] > 0.5), "case %d: is_valid" % case
        assert np.array_equal(inside[fin].astype(bool), pr[fin, 2] > 0.5), "case %d: is_inside" % case
        assert 0.05 < valid.mean() < 1.0


@pytest.mark.parametrize("setname,C,sizes", [("default34", 34, (None, 1, 3, 10)), ("yaml66", 66, (None, 1, 2, 10))])
def test_fuzz_stencils_vs_reference_source(setname, C, sizes):
    """dilation_filter_kernel (custom_kernels.py:392-449) and normal_filter_kernel (:452-506) on random planes whose masks run from
    empty to full (a hole next to nothing, a hole whose only source sits d cells away on the last anti-diagonal the kernel scans,
    no hole at all), with huge, tiny, negative and signed-zero values: the dilated plane and its mask EXACT for every compiled radius
    (None = the parameter set's own dilation_size), the normals within the 1e-6 the golden fixture uses (the reference normalises with
    a reciprocal square root)."""
    rk = _ref(setname)
    rng = np.random.default_rng(C)
    own = int(build_ref.PREBUILD[setname]["dilation_size"])
    for case, density in enumerate((0.0, 0.002, 0.02, 0.1, 0.35, 0.8, 1.0, 0.05)):
        plane = rng.uniform(-3, 3, (C, C)).astype(np.float32)
        plane[rng.uniform(0, 1, (C, C)) < 0.05] *= np.float32(1.0e6)
        plane[rng.uniform(0, 1, (C, C)) < 0.05] *= np.float32(1.0e-12)
        plane[rng.uniform(0, 1, (C, C)) < 0.02] = np.float32(-0.0)
        mask = (rng.uniform(0, 1, (C, C)) < density).astype(np.float32)
        if case == 7:
            mask *= rng.choice([0.3, 0.5, 0.51, 1.0, 2.0], (C, C)).astype(np.float32)      # the kernels test `> 0.5`, not `!= 0`
        for d in sizes:
            out = np.zeros((C, C), np.float32); om = np.zeros((C, C), np.float32)
            rk.dilation_filter(plane.copy(), mask.copy(), out, om, size=d)
            o2, m2 = eo.dilate_plane(C, own if d is None else d, plane, mask)
            assert out.tobytes() == o2.tobytes(), "%s case %d d=%s: dilated plane" % (setname, case, d)
            assert np.array_equal(om, m2), "%s case %d d=%s: dilated mask" % (setname, case, d)
        nrm = np.zeros((3, C, C), np.float32)
        rk.normal_filter(plane.copy(), mask.copy(), nrm)
        o = eo.OracleMap(eo.make_params(eo.DEFAULTS, cell_n=C))
        o.traversability_input[...] = plane
        o.elevation_map[2] = mask
        o.normals()
        ok = np.isfinite(nrm) & np.isfinite(o.normal_map)
        assert np.array_equal(np.isfinite(nrm), np.isfinite(o.normal_map)), "%s case %d: finiteness of the normals" % (setname, case)
        assert np.allclose(o.normal_map[ok], nrm[ok], atol=1e-6), "%s case %d: normals" % (setname, case)


def test_fuzz_fresh_frames_rays_off_vs_reference_source(weights):
    """the race-free fixture of test_fuse_sums_without_outliers_rays_off under six random sensor poses and cloud heights: on a fresh
    map with the visibility pass off the sequential reference run and the two-phase contract must agree in every plane -- flags, time,
    upper bound and traversability exactly, fused height and variance within the 1e-5 north_star asks for (the reference sums floats in
    point order) -- through count, fusion and average_map_kernel (custom_kernels.py:160-197, 280-389)."""
    rk = _ref("yaml202_norays")
    C, N = 202, 30000
    cfg = dict(eo.YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
    rng = np.random.default_rng(77)
    for case in range(6):
        R, t = _random_pose(rng)
        Rf = R.ravel().copy()
        om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
        p = fx.cloud(C, N, 200 + case, dz=float(rng.uniform(-0.4, 0.2)))
        p[rng.integers(0, N, 300), :2] *= np.float32(2.5)
        m = om.elevation_map.copy(); nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
        err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32); pr = p.copy()
        rk.error_counting(m, pr, Rf, t.copy(), nm, err, cnt); rk.add_points(Rf, t.copy(), nrm, pr, m, nm); rk.average_map(nm, m)
        om.count(p, R, t); om.gate(0, 0); om.fuse(p, R, t); om.commit(); om.average()
        assert int((m[2] > 0.5).sum()) > 500, "case %d: the pose must leave points on the map" % case
        for pl in (2, 3, 4, 5, 6):
            assert np.array_equal(om.elevation_map[pl], m[pl]), "case %d plane %d" % (case, pl)
        assert np.allclose(om.elevation_map[0], m[0], atol=1e-5, rtol=1e-5), "case %d: height" % case
        assert np.allclose(om.elevation_map[1], m[1], atol=1e-5, rtol=1e-5), "case %d: variance" % case


def test_fuzz_warm_frames_with_rays_differ_only_where_the_reference_races(weights):
    """test_warm_frame_race_exposed_planes_differ_only_where_the_reference_races under four random sensor poses: a warm, aged map,
    visibility pass ON -- wherever the sequential run of the reference's kernels and the two-phase contract disagree, the cell is in
    one of the reference's race classes (multi-point cell, outlier, stale cell fused this frame, cell written by the ray pass)."""
    rk = _ref("yaml202")
    C, N = 202, 30000
    rng = np.random.default_rng(99)
    for case in range(4):
        R, t = _random_pose(rng)
        Rf = R.ravel().copy()
        om = eo.OracleMap(eo.make_params(eo.YAML, cell_n=C, weights=weights))
        om.update_map_with_kernel(fx.cloud(C, 50000, 300 + case), R, t)
        for _ in range(10):
            om.update_time()
        om.update_variance()
        p = fx.cloud(C, N, 400 + case, dz=float(rng.uniform(-0.15, 0.0)))
        pre = om.elevation_map.copy()
        m = pre.copy(); nm = np.zeros((7, C, C), np.float32); nrm = om.normal_map.copy()
        err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32); pr = p.copy()
        rk.error_counting(m, pr, Rf, t.copy(), nm, err, cnt); rk.add_points(Rf, t.copy(), nrm, pr, m, nm); rk.average_map(nm, m)
        om.count(p, R, t); om.gate(0, 0); om.fuse(p, R, t); om.commit(); om.rays(p, R, t)
        racy = (om.last["n_pts"] > 1) | (om.last["n_out"] > 0) | (om.last["ray_hits"] > 0) | np.isfinite(om.last["ray_upper"])
        racy |= (pre[2] > 0.5) & (pre[4] >= 0.5) & (om.last["n_pts"] > 0)
        assert 0.0 < racy.mean() < 0.95, "case %d" % case
        om.average()
        for pl in range(7):
            diff = ~np.isclose(om.elevation_map[pl], m[pl], atol=1e-5, rtol=1e-5)
            assert not (diff & ~racy).any(), "case %d: plane %d differs on %d race-free cells" % (case, pl, int((diff & ~racy).sum()))


def test_fuzz_camera_path_vs_reference_source(weights):
    """image_to_map_correspondence_kernel + the three samplers (custom_image_kernels.py:9-271) under eight random cameras: position
    over / beside / outside the map, tilt up to grazing, narrow and wide lenses, with and without radtan distortion, rough relief
    (occlusions), map centres off the origin -- uv, validity and the sampled layers exact."""
    rk = _ref("image98")
    C = 98
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    rng = np.random.default_rng(4242)
    R0, t0 = fx.POSES["identity"]
    seen = 0
    for case in range(8):
        om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
        p = fx.cloud(C, 20000, 500 + case)
        p[:, 2] += np.float32(rng.uniform(0.1, 0.5)) * np.sin(p[:, 0] * np.float32(rng.uniform(1, 4))) * np.cos(p[:, 1] * np.float32(rng.uniform(1, 4)))
        om.update_map_with_kernel(p, R0, t0)
        H, W = int(rng.choice([32, 48, 96])), int(rng.choice([40, 64, 128]))
        f = float(rng.uniform(15.0, 120.0))
        K = np.array([[f, 0, W / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), H / 2 + rng.uniform(-3, 3)], [0, 0, 1]], np.float32)
        D = (rng.normal(0, 1, 5) * np.array([0.08, 0.02, 0.004, 0.004, 0.001])).astype(np.float32) if case % 2 else np.zeros(5, np.float32)
        Rwc = fx.rot(np.pi + rng.uniform(-1.2, 1.2), rng.uniform(-0.8, 0.8), rng.uniform(-3, 3)).astype(np.float32)
        cam = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 2.5), rng.uniform(0.6, 3.0)], np.float32)
        t = (-Rwc @ cam).astype(np.float32)
        center = rng.uniform(-0.4, 0.4, 3).astype(np.float32)
        Pm, x1, y1, z1 = fx.camera_inputs(center, C, 0.04, K, Rwc, t)
        uv_r = np.zeros((2, C, C), np.float32); va_r = np.zeros((C, C), np.bool_)
        rk.image_correspondence(om.elevation_map.copy(), x1, y1, z1, Pm.ravel().copy(), K.ravel().copy(), D.copy(), H, W, center, uv_r, va_r)
        uv, va = eo.image_correspondence(om.P, om.elevation_map, x1, y1, z1, Pm.ravel(), K.ravel(), D, H, W, center)
        assert np.array_equal(va.astype(bool), va_r), "case %d: validity" % case
        assert np.array_equal(uv, uv_r), "case %d: uv" % case
        seen += int(va_r.sum())
        img = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        rgb = rng.integers(0, 256, (3, H, W)).astype(np.float32)
        sem = rng.uniform(0, 1, (3, C, C)).astype(np.float32)
        new_r = np.zeros_like(sem)
        rk.image_fuse("exponential", sem.copy(), 0, img[1].copy(), uv_r, va_r, H, W, new_r)
        rk.image_fuse("color", sem.copy(), 1, rgb.copy(), uv_r, va_r, H, W, new_r)
        rk.image_fuse("average", sem.copy(), 2, img[2].copy(), uv_r, va_r, H, W, new_r)
        mine = sem.copy()
        eo.image_fuse(om.P, "exponential", mine[0], img[1], uv, va, H, W, 0.7)
        eo.image_fuse(om.P, "color", mine[1], rgb, uv, va, H, W)
        eo.image_fuse(om.P, "average", mine[2], img[2], uv, va, H, W)
        assert np.array_equal(mine[0], new_r[0]) and np.array_equal(mine[1].view(np.uint32), new_r[1].view(np.uint32)) and np.array_equal(mine[2], new_r[2]), "case %d" % case
    assert seen > 2000, "the sweep must see the map"


def test_fuzz_polygon_mask_vs_reference_source():
    """polygon_mask_kernel (custom_kernels.py:509-651) on 40 random polygons: 3 ... 9 vertices in random order (self-intersecting ones
    included -- the kernel's crossing test does not care), sizes from sub-cell to beyond the map, random map centres: exact."""
    params = build_ref.PREBUILD["polygon130"]
    if not ref_kernels.available(params):
        pytest.skip("compiled reference not built")
    rk = ref_kernels.RefKernels(params, build=False)
    C = 130
    P = eo.make_params(eo.DEFAULTS, cell_n=C)
    rng = np.random.default_rng(31)
    covered = 0
    for case in range(40):
        n = int(rng.integers(3, 10))
        scale = float(10.0 ** rng.uniform(-2.0, 0.6))
        center = (float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.6, 0.6)))
        poly = (rng.uniform(-1, 1, (n, 2)) * scale + rng.uniform(-1.0, 1.0, 2) + np.array(center)).astype(np.float32)
        # get_polygon_traversability clips the vertices to the map first (elevation_mapping.py:851-855)
        half = (C - 2) * 0.04 / 2
        poly = np.clip(poly, np.array(center, np.float32) - np.float32(half), np.array(center, np.float32) + np.float32(half)).astype(np.float32)
        want = np.full((C, C), -1, np.float32)
        bbox = np.concatenate([poly.min(axis=0), poly.max(axis=0)]).astype(np.float32)
        rk.polygon_mask(poly, center[0], center[1], bbox, want)
        got = eo.polygon_mask(P, poly, center[0], center[1])
        assert np.array_equal(got, want), "case %d (%d vertices, scale %.3g)" % (case, n, scale)
        covered += int((want > 0.5).sum())
    assert covered > 5000


def test_fuzz_semantic_point_fusions_vs_reference_source():
    """sum / average, sum / class_average and add_color / color_average (custom_semantic_kernels.py:9-51,167-194,233-267,270-375) driven
    as tests/golden/make_golden.py drives them, under five random poses, cloud seeds and previous layer contents: the averaged layers
    within 1e-6 (the reference adds floats in point order), the packed colour layer bit for bit."""
    rk = _ref("yaml66")
    params = build_ref.PREBUILD["yaml66"]
    C, N, K = 66, 6000, 4
    rng = np.random.default_rng(606)
    i32 = lambda *a: np.array(a, np.int32)      # noqa: E731
    for case in range(5):
        R, t = _random_pose(rng)
        t[:2] *= np.float32(0.3)                 # (a 66-cell map is 2.6 m wide)
        Rf = R.ravel().copy()
        p = fx.semantic_cloud(C, N, 700 + case)
        prev = (rng.uniform(0, 1, (C, C)) * (rng.uniform(0, 1, (C, C)) < rng.uniform(0.1, 0.9))).astype(np.float32)
        m = np.zeros((7, C, C), np.float32); m[1] = params["initial_variance"]; m[3] = 1
        nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
        err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
        xyz = np.ascontiguousarray(p[:, :3])
        rk.error_counting(m, xyz, Rf, t.copy(), nm, err, cnt); rk.add_points(Rf, t.copy(), nrm, xyz, m, nm)
        pc = p.copy(); pc[:, :3] = xyz
        sem = np.zeros((4, C, C), np.float32); sem[2] = prev
        newmap = np.zeros((4, C, C), np.float32)
        rk.sem_sum(pc, Rf, t.copy(), i32(3, 4), i32(0, 1), i32(3 + K, 2), sem, newmap, N * 2)
        rk.sem_average(newmap, i32(3, 4), i32(0, 1), i32(3 + K, 2), nm, sem, C * C * 2)
        newmap2 = np.zeros((4, C, C), np.float32)
        rk.sem_sum(pc, Rf, t.copy(), i32(5), i32(2), i32(3 + K, 1), sem, newmap2, N)
        rk.sem_class_average(newmap2, i32(5), i32(2), i32(3 + K, 1), nm, sem, C * C)
        color_map = np.zeros((4, C, C), np.uint32)
        rk.sem_add_color(pc, Rf, t.copy(), i32(6), i32(3), i32(3 + K, 1), color_map, N)
        rk.sem_color_average(color_map, i32(6), i32(3), i32(3 + K, 1), sem, C * C)
        om = eo.OracleMap(eo.make_params(eo.YAML, cell_n=C))
        om.count(p, R, t); om.gate(0, 0); om.fuse(p, R, t)
        assert np.array_equal(om.last["cnt"], nm[2].astype(np.uint32)), "case %d: accepted points per cell" % case
        assert int((nm[2] > 1).sum()) > 100, "case %d: cells with several points" % case
        om.semantic_map = np.zeros((4, C, C), np.float32); om.semantic_map[2] = prev
        om.semantic_update(p, R, t, average=[(3, 0), (4, 1)], class_average=[(5, 2)], color=[(6, 3)], alpha=0.5)
        assert np.allclose(om.semantic_map[:3], sem[:3], atol=1e-6, rtol=1e-6), "case %d: averaged layers" % case
        assert np.array_equal(om.semantic_map[3].view(np.uint32), sem[3].view(np.uint32)), "case %d: packed colour" % case


def test_fuzz_bayesian_point_fusions_vs_reference_source():
    """class_bayesian (alpha_kernel + renormalisation, the K = 2 launch-size quirk, negative theta ignored, persistent pseudo-counts)
    and bayesian_inference (sum_compact + bayesian_inference kernels) as the reference's fusions run them
    (fusion/pointcloud_class_bayesian.py:56-75, fusion/pointcloud_bayesian_inference.py:101-122; recipe of tests/golden/make_golden.py),
    under five random poses, clouds, priors and previous layers."""
    rk = _ref("bayes66")
    params = build_ref.PREBUILD["bayes66"]
    C, N, K = 66, 6000, 4
    rng = np.random.default_rng(808)
    i32 = lambda *a: np.array(a, np.int32)      # noqa: E731
    for case in range(5):
        R, t = _random_pose(rng)
        t[:2] *= np.float32(0.3)
        Rf = R.ravel().copy()
        p = fx.bayes_cloud(C, N, 900 + case)
        prior = rng.uniform(0, 2, (2, C, C)).astype(np.float32); prior[:, rng.uniform(0, 1, (C, C)) < rng.uniform(0.1, 0.6)] = 0.0
        prev = (rng.uniform(0, 1, (C, C)) * (rng.uniform(0, 1, (C, C)) < 0.5)).astype(np.float32)
        m = np.zeros((7, C, C), np.float32); m[1] = params["initial_variance"]; m[3] = 1
        nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
        err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
        xyz = np.ascontiguousarray(p[:, :3])
        rk.error_counting(m, xyz, Rf, t.copy(), nm, err, cnt); rk.add_points(Rf, t.copy(), nrm, xyz, m, nm)
        pc = p.copy(); pc[:, :3] = xyz
        sem = np.zeros((3, C, C), np.float32); sem[2] = prev
        newmap = np.zeros((3, C, C), np.float32); newmap[:2] = prior
        rk.alpha(pc, i32(3, 4), i32(0, 1), i32(3 + K, 2), newmap, N)
        sum_alpha = np.sum(newmap[[0, 1]], axis=0); sum_alpha[sum_alpha == 0] = 1
        sem[[0, 1]] = newmap[[0, 1]] / np.expand_dims(sum_alpha, axis=0)
        sum_mean = np.zeros((1, C, C), np.float32)
        rk.sum_compact(pc, Rf, t.copy(), i32(5), i32(2), i32(3 + K, 1), sum_mean, N)
        rk.bayesian_inference(i32(5), i32(2), i32(3 + K, 1), nm, newmap, sum_mean, sem, C * C)
        om = eo.OracleMap(eo.make_params(eo.YAML, cell_n=C))
        om.count(p, R, t); om.gate(0, 0); om.fuse(p, R, t)
        om.semantic_map = np.zeros((3, C, C), np.float32); om.semantic_map[2] = prev
        om.semantic_alpha = np.zeros((3, C, C), np.float32); om.semantic_alpha[:2] = prior
        om.semantic_update(p, R, t, class_bayesian=[(3, 0), (4, 1)], bayesian_inference=[(5, 2)])
        assert np.allclose(om.semantic_alpha[:2], newmap[:2], atol=1e-5, rtol=1e-5), "case %d: pseudo-counts" % case
        assert np.allclose(om.semantic_map[:2], sem[:2], atol=1e-6, rtol=1e-5), "case %d: class probabilities" % case
        assert np.array_equal(om.semantic_map[2], sem[2]), "case %d: bayesian_inference layer" % case
        assert (newmap[:2] != prior).any()
