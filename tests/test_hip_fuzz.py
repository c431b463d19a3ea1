"""GPU: seeded random scenarios, HIP frames against the oracle BIT FOR BIT.  The fixed parity cases exercise the configurations of
BASELINE.json; this sweep walks the space between them -- map sizes that are not multiples of the 16 x 64 tiles, both index modes,
every feature toggle, both scatter paths (and stacked bins), clouds with NaN rows / far outliers / piled-up cells, sensors off
centre, frames interleaved with decay passes and xy map moves.  Every scenario is a pure function of its index: a failure
reproduces with `-k "fuzz and <index>"`."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_equal, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu

SIZES = [50, 66, 98, 127, 130, 202, 257, 300]


def _scenario(k):
    rng = np.random.default_rng(7000 + k)
    cfg = dict(eo.YAML if rng.random() < 0.6 else eo.DEFAULTS)
    for key in ("enable_visibility_cleanup", "enable_overlap_clearance", "enable_edge_sharpen", "enable_drift_compensation"):
        cfg[key] = bool(rng.random() < 0.7)
    cfg["dilation_size"] = int(rng.integers(1, 4))
    if rng.random() < 0.3:
        cfg["wall_num_thresh"] = int(rng.integers(1, 6))              # edge sharpening / wall test actually fire
    if rng.random() < 0.3:
        cfg["max_ray_length"] = float(rng.choice([1.0, 3.0, 6.0]))
    C = int(rng.choice(SIZES))
    mode = "reference_fp16" if rng.random() < 0.6 else "fp32"
    scatter = [("atomic", 0), ("binned", 0), ("binned", 2), ("binned", 4), ("auto", 0)][int(rng.integers(0, 5))]
    return rng, cfg, C, mode, scatter


def _pose(rng):
    a = rng.uniform(-0.5, 0.5, 3)
    R = fx.rot(*a) if rng.random() < 0.7 else np.eye(3, dtype=np.float32)
    t = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.6, 1.4)], np.float32)
    return R.astype(np.float32), t


def _cloud(rng, C, N, seed):
    p = fx.cloud(C, N, seed, dz=float(rng.uniform(-0.25, 0.1)))
    if rng.random() < 0.5:
        m = min(p[::3].shape[0], p[1::3].shape[0])
        p[::3][:m, :2] = p[1::3][:m, :2]                               # several points per cell
    if rng.random() < 0.5:
        p[::211, int(rng.integers(0, 3))] = np.nan
    if rng.random() < 0.5:
        p[7::173] *= float(rng.uniform(1.5, 4.0))                      # outside the map / beyond the height gates
    if rng.random() < 0.3:
        p[:, 2] += 0.2 * np.sin(p[:, 0] * 3.0) * np.cos(p[:, 1] * 2.0)  # relief: walls, occluded cells
    return p


@pytest.mark.parametrize("k", range(40))
def test_fuzz_frames_bitwise(k, weights):
    rng, cfg, C, mode, (scatter, stack) = _scenario(k)
    hip, orc = make_pair(cfg, C, mode, weights)
    hip.set_scatter_mode(scatter, bin_stack=stack)
    orc.center = np.zeros(3, np.float32)
    desc = "scenario %d: C=%d %s %s/%d rays=%s overlap=%s edge=%s drift=%s d=%d" % (
        k, C, mode, scatter, stack, cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"], cfg["enable_edge_sharpen"],
        cfg["enable_drift_compensation"], cfg["dilation_size"])
    for f in range(3):
        R, t = _pose(rng)
        N = int(rng.integers(2000, 60000)) if rng.random() < 0.85 else int(rng.integers(131072, 180000))      # (>= 131072 points: scatter mode 'auto' would sort)
        p = _cloud(rng, C, N, 100 * k + f)
        pn, on = (1.0, 1.0) if rng.random() < 0.6 else (0.0, 0.0)
        tw = (t + hip.center).astype(np.float32)
        hip.update_map_with_kernel(p, [], R, tw.copy(), pn, on)
        orc.update_map_with_kernel(p, R, (tw - orc.center).astype(np.float32), pn, on)
        what = "%s, frame %d (N=%d, noise %g)" % (desc, f, N, pn)
        assert_planes_equal(hip.elevation_map, orc.elevation_map, what=what)
        assert_planes_equal(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"], what=what)
        assert np.array_equal(hip.traversability_input.view(np.uint32), orc.traversability_input.view(np.uint32)), what + ": dilated plane"
        for _ in range(int(rng.integers(0, 13))):
            hip.update_time(); orc.update_time()
        if rng.random() < 0.5:
            hip.update_variance(); orc.update_variance()
        if rng.random() < 0.6:                                          # xy move by whole and fractional cells (z stays: float32 vs float64 offsets)
            v = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 0.0])
            hip.move(v); orc.move(v)
            assert np.array_equal(hip.center, orc.center), what
        assert_planes_equal(hip.elevation_map, orc.elevation_map, what=what + " after decay / move")


@pytest.mark.parametrize("k", range(16))
def test_fuzz_semantic_channels(k, weights):
    """random channel layouts of a multi-modal cloud (1 .. 7 extra columns; average / class_average / at most one colour channel in
    any order, so the tile kernel's group loop, its one-load gather of adjacent columns and the colour channel riding along are all
    hit), both scatter paths, two frames: averaged layers within 1e-6, the packed colour layer exact."""
    rng = np.random.default_rng(8200 + k)
    C = int(rng.choice([66, 98, 130, 202]))
    K = int(rng.integers(1, 8))
    kinds = [str(rng.choice(["average", "class_average"])) for _ in range(K)]
    if rng.random() < 0.7:
        kinds[int(rng.integers(0, K))] = "color"
    names = ["ch%d" % j for j in range(K)]
    mode = "reference_fp16" if rng.random() < 0.6 else "fp32"
    hip, orc = make_pair(dict(eo.YAML, enable_visibility_cleanup=bool(rng.random() < 0.5)), C, mode, weights)
    hip.param.pointcloud_channel_fusions = {n: kd for n, kd in zip(names, kinds)}
    hip.set_scatter_mode("binned" if rng.random() < 0.6 else "atomic")
    R, t = _pose(rng)
    groups = {kd: [(3 + j, j) for j in range(K) if kinds[j] == kd] for kd in ("average", "class_average", "color")}
    for f in range(2):
        N = int(rng.integers(4000, 50000))
        p = fx.cloud(C, N, 300 * k + f, extra=K)
        m = min(p[::3].shape[0], p[1::3].shape[0])
        p[::3][:m, :2] = p[1::3][:m, :2]
        for j in range(K):
            if kinds[j] == "color":
                p[:, 3 + j] = rng.integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
        hip.input_pointcloud(p, ["x", "y", "z"] + names, R, t.copy(), 0.0, 0.0)
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, average=groups["average"], class_average=groups["class_average"], color=groups["color"], alpha=0.5, n_layers=K)
        hip.update_time(); orc.update_time()
        sm = hip.semantic_map.semantic_map
        what = "scenario %d (C=%d, kinds=%s, %s), frame %d" % (k, C, kinds, mode, f)
        assert hip.semantic_map.layer_names == names
        for j in range(K):
            if kinds[j] == "color":
                assert np.array_equal(sm[j].view(np.uint32), orc.semantic_map[j].view(np.uint32)), what + ": colour layer %d" % j
            else:
                assert np.allclose(sm[j], orc.semantic_map[j], atol=1e-6, rtol=1e-5), what + ": layer %d" % j
        assert_planes_equal(hip.elevation_map, orc.elevation_map, what=what)


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_edge_cloud_sizes(scatter, dtype, weights):
    """clouds of 0, 1, 63 .. 4097 points, all-NaN and all-outside clouds, growing and shrinking between frames (buffer regrowth of
    the upload path and of the sort), float32 and float64 host clouds through input_pointcloud: every frame bit for bit"""
    C = 130
    hip, orc = make_pair(dict(eo.YAML), C, "reference_fp16", weights)
    hip.set_scatter_mode(scatter)
    R, t = fx.POSES["rotated"]
    sizes = [0, 1, 63, 64, 65, 4095, 4096, 4097, 20000, 2, 0, 9000]
    for f, N in enumerate(sizes):
        p = fx.cloud(C, N, 400 + f, dz=-0.01 * f)
        if f == 5:
            p[:] = np.nan                                               # nothing survives the NaN filter (elevation_mapping.py:458)
        if f == 7:
            p[:, :2] *= 10.0                                            # every point outside the map
        hip.input_pointcloud(p.astype(dtype), ["x", "y", "z"], R, t.copy(), 1.0, 1.0)
        pv = p[~np.isnan(p).any(axis=1)]                                # the oracle's frame takes the filtered cloud, like update_map_with_kernel
        orc.update_map_with_kernel(pv, R, t, 1.0, 1.0)
        hip.update_time(); orc.update_time()
        what = "%s %s frame %d (N=%d)" % (scatter, np.dtype(dtype).name, f, N)
        assert_planes_equal(hip.elevation_map, orc.elevation_map, what=what)
        assert_planes_equal(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"], what=what)
