"""Race-free warm fixtures shared by the CPU and GPU tests and by tests/golden/make_golden.py.

ONE point per frame on an injected warm map: with a single point the reference kernels cannot race with themselves, so their
sequential execution (oracle/_ref) IS the contract -- including the two branches that only a warm map reaches:
the outlier variance inflation (reference custom_kernels.py:173-175) and the ray penetration (:236-258: the
`-cleanup_step / (ray_length / max_ray_length)` decrement, the cosine test against the float16-rounded normal, the
`wall_num_thresh && time < 1.0` skip).  The `wall3` fixture reaches that last skip with THREE points per frame (two edge-skipped
inliers in one cell, a third ray crossing it) under a parameter set with wall_num_thresh = 1; its frames are kept only if the
compiled reference gives the same planes for the forward and the reversed point order (race free by construction of the data)."""
import numpy as np

import _fixtures as fx

K_FRAMES = 400
SETS = {"yaml202": "YAML", "default202": "DEFAULTS"}


def base_after_reset(m, init_var):
    """average_map_kernel resets every cell with is_valid < 0.5 each frame (reference custom_kernels.py:380-384)"""
    b = m.copy()
    inv = b[2] < 0.5
    b[0][inv] = 0.0; b[1][inv] = init_var; b[2][inv] = 0.0
    return b


def sparse_diff(base, final):
    idx = np.flatnonzero(base.view(np.uint32).ravel() != final.view(np.uint32).ravel()).astype(np.int32)
    return idx, final.ravel()[idx].copy()


def apply_sparse(base, idx, val):
    out = base.copy()
    out.ravel()[idx] = val
    return out


def ref_frame(rk, m, nrm, p, R, t):
    """error_counting -> add_points -> average_map of the compiled reference, in place on m (elevation_mapping.py:334-369)"""
    C = m.shape[1]
    nm = np.zeros((7, C, C), np.float32); err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
    pr = np.ascontiguousarray(p[:, :3]).copy()
    Rf = np.ascontiguousarray(R, np.float32).ravel().copy(); tf = np.ascontiguousarray(t, np.float32).copy()
    rk.error_counting(m, pr, Rf, tf, nm, err, cnt)
    rk.add_points(Rf, tf, nrm, pr, m, nm)
    rk.average_map(nm, m)
    return nm


def oracle_frame(om, p, R, t):
    """the same three kernels of the contract oracle (count -> gate(closed) -> fuse -> commit -> rays -> average)"""
    om.count(p, R, t); om.gate(0.0, 0.0); om.fuse(p, R, t); om.commit()
    if om.P.enable_visibility_cleanup:
        om.rays(p, R, t)
    hits = int(om.last["ray_hits"].sum()) if "ray_hits" in om.last else 0
    outl = int(om.last["n_out"].sum())
    om.average()
    return hits, outl


def hip_frame(hip, p, R, t):
    hip.bind_points(p)
    hip.stage("count", R, t); hip.stage("gate", position_noise=0.0, orientation_noise=0.0)
    hip.stage("fuse", R, t); hip.stage("commit")
    if hip.param.enable_visibility_cleanup:
        hip.stage("rays", R, t)
    hip.stage("average")


WALL_CFG = dict(wall_num_thresh=1)        # on top of the Parameter defaults; compiled reference: build_ref.PREBUILD["wall202"]
WALL_FRAMES = 120


def wall_sequence(C, K=WALL_FRAMES, res=0.04):
    """K three-point frames for the wall-skip branch (reference custom_kernels.py:246-247: `newmap[3][cell] > wall_num_thresh
    && time < 1.0`), pose "identity".  Per frame a cell B 0.8-1.2 m from the sensor is injected with a confident height, a normal
    along the crossing ray and time in {0.9, 0.6} (inside the wall window); points A, A' fall into B with a height inside the
    edge-sharpening window (the fusion skips them, :177-179, so B keeps its time), point C lies behind B and lower, so its ray
    passes below B's height.  Even frames: B is traversable, A and A' are drift inliers => 2 > wall_num_thresh = 1 => B is
    SKIPPED; odd frames (control): B's traversability is below traversability_inlier, no inliers are counted => B is PENETRATED.
    Either way nothing a point reads is written by another point's ray, so the frames are race free (the generator checks
    forward == reversed point order on the compiled reference).  Yields (ix, iy, cell (7,), normal (3,), points (3,3), skipped)."""
    rng = np.random.default_rng(9002)
    for k in range(K):
        ang = rng.uniform(0, 2 * np.pi); dist = rng.uniform(0.8, 1.2)
        ix, iy = int(dist * np.cos(ang) / res + C / 2), int(dist * np.sin(ang) / res + C / 2)
        cx, cy = (ix + 0.5 - C / 2) * res, (iy + 0.5 - C / 2) * res
        hB = rng.uniform(0.75, 0.95); vB = rng.uniform(0.015, 0.04); tB = [0.9, 0.9, 0.6, 0.6][k % 4]
        skipped = k % 2 == 0
        d3 = np.array([cx, cy, hB - 0.12 - 1.0]); d3 /= np.linalg.norm(d3)
        f = rng.uniform(1.5, 1.7)
        A = np.array([cx, cy, hB - 1.8 * vB - 1.0], np.float32)
        Cc = np.array([cx * f, cy * f, f * (hB - 0.12 - 1.0)], np.float32)
        pts = np.stack([A, A + np.float32([0.003, -0.004, 0.0]), Cc]).astype(np.float32)
        cell = np.array([hB, vB, 1.0, 0.9 if skipped else 0.05, tB, hB, 0.0], np.float32)
        yield ix, iy, cell, d3.astype(np.float32), pts, skipped


def run_single(cfg, C, seed, frame_fn, state, tick):
    """drives K_FRAMES one-point frames; `frame_fn(p, R, t)` runs one frame on `state`, `tick()` advances the time plane"""
    for k, (p, pose) in enumerate(fx.single_points(C, K_FRAMES, seed)):
        R, t = fx.POSES[pose]
        frame_fn(p, R, t)
        if k % 10 == 0:                      # fused cells become stale again, so later rays can act on them
            tick()


def base_single(m0, cfg):
    """what the injected map looks like after K_FRAMES frames that touch nothing: unknown cells reset, time ticked"""
    b = base_after_reset(m0, np.float32(cfg["initial_variance"]))
    for k in range(0, K_FRAMES, 10):
        b[4] += np.float32(cfg["time_interval"])
    return b
