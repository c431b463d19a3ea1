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
