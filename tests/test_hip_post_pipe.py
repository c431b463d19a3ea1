"""GPU: k_post_pipe (persistent workgroups, the next tile's loads in flight; round 5) leaves the SAME planes as k_post.

The pipelined kernel is selected by map size (many tiles per workgroup: 4096^2 and up, where tests/test_hip_large_maps.py ties it to the
oracle).  Here it is forced onto small maps with a small grid (EMAP_POST_PIPE=1, EMAP_POST_PIPE_GRID: both read at every launch) so that
every workgroup walks a dozen tiles -- interior tiles (prefetched), edge tiles (staged by the general walk), tiles with and without
holes, mostly-unknown tiles (reach masks), map sides that are no multiple of the tile, a shifted circular origin -- and compared bit for
bit with the plain kernel AND with the oracle.  Reference: dilation_filter_kernel / normal_filter_kernel (custom_kernels.py:392-506),
traversability_filter.py:8-47."""
import os

import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_equal, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _oracle_threads():
    eo.set_threads(16)
    yield
    eo.set_threads(1)
    os.environ.pop("EMAP_POST_PIPE", None)
    os.environ.pop("EMAP_POST_PIPE_GRID", None)


def _pipe(on, grid=40):
    os.environ["EMAP_POST_PIPE"] = "1" if on else "0"
    os.environ["EMAP_POST_PIPE_GRID"] = str(grid)


def _stencil_outputs(hip):
    return hip.normal_map.copy(), hip.get_layer_raw(3).copy(), np.asarray(hip.traversability_input).copy()


@pytest.mark.parametrize("C,grid", [(1024, 40), (1000, 37), (1100, 512)])
def test_pipelined_stencils_equal_plain_and_oracle(C, grid, weights):
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    ref, _ = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    for f, (n, dz) in enumerate(((400_000, 0.0), (150_000, -0.04), (900_000, 0.03))):
        p = fx.cloud(C, n, 20 + f, dz=dz)
        _pipe(True, grid)
        hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        _pipe(False)
        ref.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        for a, b, what in zip(_stencil_outputs(hip), _stencil_outputs(ref), ("normals", "traversability", "traversability_input")):
            assert a.tobytes() == b.tobytes(), "frame %d: %s differ between k_post_pipe and k_post (%d cells)" % (f, what, int((a != b).sum()))
        assert_planes_equal(hip.elevation_map, orc.elevation_map, what="frame %d" % f)
        assert_planes_equal(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"], what="frame %d normals" % f)
        for _ in range(3):
            hip.update_time(); ref.update_time(); orc.update_time()


def test_pipelined_stencils_on_a_mostly_unknown_moving_map(weights):
    """the ray-cast terrain scene (86 % holes: reach masks, long hole searches) on a map whose circular origin moves between the frames"""
    C = 1024
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    hip, _ = make_pair(cfg, C, "reference_fp16", weights)
    ref, _ = make_pair(cfg, C, "reference_fp16", weights)
    R, t0 = fx.POSES["identity"]
    moves = [(0.0, 0.0, 0.0), (0.52, -0.36, 0.0), (-1.48, 0.8, 0.05), (0.12, 2.04, 0.0)]
    for f, mv in enumerate(moves):
        p = fx.terrain_cloud(C, 1000, 300, 5 + f, shift=0.1 * f)
        for m, on in ((hip, True), (ref, False)):
            _pipe(on, 48)
            if f:
                m.move_to(np.array(mv, np.float64), np.eye(3))
            m.update_map_with_kernel(p, [], R, (t0 + m.center).astype(np.float32), 1.0, 1.0)
        for a, b, what in zip(_stencil_outputs(hip), _stencil_outputs(ref), ("normals", "traversability", "traversability_input")):
            assert a.tobytes() == b.tobytes(), "frame %d: %s differ between k_post_pipe and k_post (%d cells)" % (f, what, int((a != b).sum()))
        assert hip.elevation_map.tobytes() == ref.elevation_map.tobytes()
    assert (hip.elevation_map[2] > 0.5).mean() < 0.5          # the scene really is mostly unknown
