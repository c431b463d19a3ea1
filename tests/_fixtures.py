"""Seed-defined inputs shared by the tests and by tests/golden/make_golden.py."""
import numpy as np


def cloud(C, N, seed, res=0.04, dz=0.0, extra=0):
    """x, y ~ U(-L/2, L/2) with L = C*res, sensor-frame z ~ U(-0.5, 0.5) + dz; float32 (N, 3+extra)."""
    rng = np.random.default_rng(seed)
    L = C * res / 2
    p = np.empty((N, 3 + extra), np.float32)
    p[:, 0] = rng.uniform(-L, L, N)
    p[:, 1] = rng.uniform(-L, L, N)
    p[:, 2] = rng.uniform(-0.5, 0.5, N) + dz
    for k in range(extra):
        p[:, 3 + k] = rng.uniform(0, 1, N)
    return p


def rot(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


POSES = {
    "identity": (np.eye(3, dtype=np.float32), np.array([0, 0, 1], np.float32)),
    "rotated": (rot(np.deg2rad(10), np.deg2rad(20), np.deg2rad(30)), np.array([0.3, -0.2, 1.1], np.float32)),
}


def stencil_inputs(C, seed):
    rng = np.random.default_rng(seed)
    plane = rng.uniform(-1, 1, (C, C)).astype(np.float32)
    mask = (rng.uniform(0, 1, (C, C)) < 0.35).astype(np.float32) + (rng.uniform(0, 1, (C, C)) < 0.1).astype(np.float32)
    return plane, mask.astype(np.float32)


def semantic_cloud(C, N, seed):
    """(N, 7): x y z | two features U(0,1) | one class probability U(0,1) | packed 0x00RRGGBB bit-cast to float32
    (wire format of the semantic sensor: reference sensor_processing/.../pointcloud_node.py:159-171)."""
    p = cloud(C, N, seed, extra=4)
    rng = np.random.default_rng(1000 + seed)
    rgb = rng.integers(0, 1 << 24, N, dtype=np.uint32)
    p[:, 6] = rgb.view(np.float32)
    p[::3, :2] = p[1::3, :2][: p[::3].shape[0]]      # pile points up so that cells see several points
    return p


def bayes_cloud(C, N, seed):
    """semantic_cloud whose feature columns 3..5 also hold negative values (class_bayesian ignores theta < 0)"""
    p = semantic_cloud(C, N, seed)
    p[:, 3:6] -= np.float32(0.2)
    return p


def bayes_alpha_prior(C):
    """pseudo-counts already accumulated in two class_bayesian layers (30 % of the cells still empty)"""
    rng = np.random.default_rng(78)
    a = rng.uniform(0, 2, (2, C, C)).astype(np.float32)
    a[:, rng.uniform(0, 1, (C, C)) < 0.3] = 0.0
    return a


def semantic_prev(C):
    rng = np.random.default_rng(77)
    prev = rng.uniform(0, 1, (C, C)).astype(np.float32)
    prev[rng.uniform(0, 1, (C, C)) < 0.5] = 0.0
    return prev


def camera_case(C, seed, with_distortion):
    """a camera above the map looking down-forward: K, D, R (world->camera), t, image size"""
    rng = np.random.default_rng(300 + seed)
    H, W = 48, 64
    K = np.array([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]], np.float32)
    D = (np.array([0.05, -0.01, 0.002, -0.001, 0.0005], np.float32) if with_distortion else np.zeros(5, np.float32))
    Rwc = rot(np.pi + 0.35 * rng.uniform(-1, 1), 0.3 * rng.uniform(-1, 1), rng.uniform(-3, 3))   # optical axis roughly -z
    cam_pos = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 1.6], np.float32)
    t = (-Rwc @ cam_pos).astype(np.float32)
    return K, D, Rwc.astype(np.float32), t, H, W


def camera_inputs(emap_center, cell_n, resolution, K, R, t):
    """P, x1, y1, z1 exactly as reference input_image computes them (elevation_mapping.py:527-534)"""
    P = (K @ np.concatenate([R, t[:, None]], 1)).astype(np.float32)
    t_cam_map = -R.T @ t - emap_center
    x1 = np.float32(np.uint32((cell_n / 2) + (t_cam_map[0] / resolution)))
    y1 = np.float32(np.uint32((cell_n / 2) + (t_cam_map[1] / resolution)))
    return P, x1, y1, np.float32(t_cam_map[2])


def warm_map(C, seed, init_var):
    """Injected warm map for the race-free single-point fixtures: random heights around the sensor height, variances over
    3.5 decades (=> outliers AND accepted points), 70 % known cells, `time` in the classes the visibility pass distinguishes
    (fresh < 0.5 <= wall window < 1.0 <= stale), random upper bounds, random unit normals.  Returns (map (7,C,C), normals (3,C,C))."""
    rng = np.random.default_rng(4000 + seed)
    m = np.zeros((7, C, C), np.float32)
    m[0] = rng.uniform(0.2, 2.2, (C, C))
    m[1] = 10.0 ** rng.uniform(-3.0, 0.5, (C, C))
    m[2] = (rng.uniform(0, 1, (C, C)) < 0.7)
    m[3] = rng.uniform(0, 1, (C, C))
    m[4] = rng.choice(np.array([0.0, 0.6, 0.9, 1.5, 3.0], np.float32), (C, C))
    m[5] = rng.uniform(0.0, 2.5, (C, C))
    m[6] = (rng.uniform(0, 1, (C, C)) < 0.5)
    inv = m[2] < 0.5                                   # unknown cells look like the reference leaves them
    m[0][inv] = 0.0; m[1][inv] = init_var
    n = rng.normal(0, 1, (3, C, C))
    n /= np.linalg.norm(n, axis=0, keepdims=True)
    return m, n.astype(np.float32)


def single_points(C, K, seed):
    """K one-point frames: (point (1,3), pose name).  Points are uniform over the map like `cloud`; poses alternate."""
    p = cloud(C, K, 7000 + seed)
    names = ["rotated", "identity"]
    return [(p[k:k + 1].copy(), names[k % 2]) for k in range(K)]


def semantic_kernel_cases():
    """Inputs of the raw semantic kernels (EM/kernels/custom_semantic_kernels.py) on the 4 x 4 toy map of the reference's own tests
    (EM/tests/test_semantic_kernels.py:25-307: three points in cell 1, all valid and inside) plus a richer draw on the same map:
    20 points over all 16 cells with some invalid / outside rows, negative class probabilities, previous layer values.
    Returns {case: dict of arrays}; points rows are (idx, valid, inside, ch0, ch1, ch2)."""
    toy = np.array([[1, 1, 1, 0.3, 0.3, 0.0], [1, 1, 1, 0.1, 0.2, 0.0], [1, 1, 1, 0.1, 0.2, 0.0]], np.float32)
    rng = np.random.default_rng(2024)
    rich = np.zeros((20, 6), np.float32)
    rich[:, 0] = rng.integers(0, 16, 20)
    rich[:, 1] = rng.uniform(0, 1, 20) < 0.85
    rich[:, 2] = rng.uniform(0, 1, 20) < 0.85
    rich[:, 3:6] = rng.uniform(-0.3, 1.0, (20, 3))
    out = {}
    for name, pts in (("toy", toy), ("rich", rich)):
        n = pts.shape[0]
        col = pts.copy()
        col[:, 3] = rng.integers(0, 1 << 24, n, dtype=np.uint32).view(np.float32)     # packed 0x00RRGGBB in channel 3 for the colour kernels
        elmap = np.zeros((3, 4, 4), np.float32)
        elmap[2] = np.bincount(pts[(pts[:, 1] != 0) & (pts[:, 2] != 0), 0].astype(int), minlength=16).reshape(4, 4)
        prev = rng.uniform(0, 1, (4, 4, 4)).astype(np.float32) * (rng.uniform(0, 1, (4, 4, 4)) < 0.6)
        out[name] = dict(points=pts, points_color=col, pcl_ids=np.array([3, 4], np.int32), layer_ids=np.array([1, 2], np.int32),
                         new_elmap=elmap, prev=prev.astype(np.float32), sigma=rng.uniform(0.1, 2.0, (4, 4, 4)).astype(np.float32),
                         max_pt=rng.uniform(0, 1, (n, 2)).astype(np.float32), max_id=rng.integers(0, 4, (n, 2)).astype(np.int32))
    return out


def semantic_kernel_run(K, c):
    """K: an object with the reference kernels' call surface (oracle/ref_kernels.RefKernels)"""
    pts, col, pc, ml, el = c["points"], c["points_color"], c["pcl_ids"], c["layer_ids"], c["new_elmap"]
    n, stride, nch = pts.shape[0], pts.shape[1], pc.shape[0]
    chn = np.array([stride, nch, 2], np.int32)
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.zeros(3, np.float32)
    out = {}
    newmap = np.zeros((4, 4, 4), np.float32); smap = c["prev"].copy()
    K.sem_sum(pts, R, t, pc, ml, chn, smap, newmap, n * nch); out["sum_newmap"] = newmap.copy()
    avg = c["prev"].copy(); K.sem_average(newmap, pc, ml, chn, el, avg, 16 * nch); out["average_map"] = avg
    cav = c["prev"].copy(); K.sem_class_average(newmap, pc, ml, chn, el, cav, 16 * nch); out["class_average_map"] = cav
    sm = np.zeros((nch, 4, 4), np.float32); K.sum_compact(pts, R, t, pc, ml, chn, sm, n * nch); out["sum_compact"] = sm.copy()
    bmap = c["prev"].copy(); sig = c["sigma"].copy()
    K.bayesian_inference(pc, ml, chn, el, sig, sm, bmap, 16 * nch); out["bayes_map"] = bmap; out["bayes_sigma"] = sig
    al = np.zeros((4, 4, 4), np.float32); K.alpha(pts, pc, ml, chn, al, n * nch); out["alpha_newmap"] = al
    mx = np.zeros((4, 4, 4), np.float32); K.sem_sum_max(pts, c["max_pt"], c["max_id"], pc, ml, chn, mx, n); out["sum_max_newmap"] = mx
    cpc, cml = np.array([3], np.int32), np.array([0], np.int32); cch = np.array([stride, 1], np.int32)
    cm = np.zeros((4, 4, 4), np.uint32); K.sem_add_color(col, R, t, cpc, cml, cch, cm, n); out["color_map"] = cm.copy()
    cs = c["prev"].copy(); K.sem_color_average(cm, cpc, cml, cch, cs, 16); out["color_average_map"] = cs
    return out


def terrain_height(x, y, boxes):
    """the scene of terrain_cloud: rolling ground (+-0.4 m) with box-shaped obstacles (walls) on it; world frame"""
    h = 0.3 * np.sin(0.35 * x) * np.cos(0.27 * y) + 0.1 * np.sin(1.3 * x + 0.5 * y)
    for (x0, x1, y0, y1, top) in boxes:
        h = np.where((x >= x0) & (x <= x1) & (y >= y0) & (y <= y1), np.maximum(h, top), h)
    return h


def terrain_boxes(C, res, shift=0.0):
    """a handful of walls and blocks, scaled with the map; `shift` moves every second one diagonally (a scene that changed between frames)"""
    L = C * res / 2
    u = L / 20.0
    raw = [(3, 3.4, -6, 6, 1.6), (-9, -5, 4, 4.3, 1.2), (-4, -2.5, -8, -6.5, 0.7), (8, 12, -12, -11.6, 1.5), (-14, -13.6, -10, 2, 0.9), (6, 7, 7, 8, 0.5)]
    out = []
    for k, (x0, x1, y0, y1, top) in enumerate(raw):
        d = shift if k % 2 else 0.0                                  # along x AND y: a wall leaves the cells it stood on
        out.append(((x0 + d) * u, (x1 + d) * u, (y0 + d) * u, (y1 + d) * u, top))
    return out


def terrain_cloud(C, n_az, n_el, seed, res=0.04, sensor_h=1.0, shift=0.0, noise=0.01):
    """A spatially coherent, SCAN-ORDERED cloud: a sensor `sensor_h` above the origin casts n_az x n_el beams (azimuth-major, elevation
    -60 .. -2.9 degrees) at terrain_height(); every beam is marched to its first intersection (coarse steps + bisection) and gets range
    noise.  Sensor frame (the map frame is R p + t with R = I, t = (0, 0, sensor_h)); float32 (n_az * n_el, 3).  Beams without a hit
    inside 1.3 x the half width end there (outside the map).  Unlike cloud() -- white noise of +-0.5 m in z -- neighbouring beams end in
    neighbouring cells, rays do not dive under the ground they measured, and walls cast shadows."""
    rng = np.random.default_rng(seed)
    L = C * res / 2
    boxes = terrain_boxes(C, res, shift)
    az = np.repeat(np.linspace(-np.pi, np.pi, n_az, endpoint=False), n_el)
    el = np.tile(np.deg2rad(np.linspace(-60.0, -2.9, n_el)), n_az)
    dx, dy, dz = np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)
    r_max = 1.3 * L
    step = max(res * 4, r_max / 160)
    lo = np.full_like(az, r_max)
    hi = np.full_like(az, r_max)
    live = np.arange(az.shape[0])                                  # beams still above the ground
    r = 0.0
    for _ in range(int(np.ceil(r_max / step))):
        r_next = r + step
        below = (sensor_h + dz[live] * r_next) <= terrain_height(dx[live] * r_next, dy[live] * r_next, boxes)
        hit = live[below]
        lo[hit] = r; hi[hit] = r_next
        live = live[~below]
        r = r_next
        if live.size == 0:
            break
    for _ in range(10):
        mid = 0.5 * (lo + hi)
        below = (sensor_h + dz * mid) <= terrain_height(dx * mid, dy * mid, boxes)
        hi = np.where(below, mid, hi); lo = np.where(below, lo, mid)
    rr = hi + rng.normal(0.0, noise, az.shape)
    return np.stack([dx * rr, dy * rr, dz * rr], axis=1).astype(np.float32)
