"""GPU: the ``pointcloud_class_max`` fusion (EM/fusion/pointcloud_class_max.py) through ``ElevationMap.input_pointcloud`` against the
NumPy restatement of the reference's statements (oracle/class_max.py): probabilities and class ids bit for bit, over several frames
(the fusion's ``unique_id`` set grows), next to an ``average`` channel in the same cloud, and across a map move (the id planes move
with the map)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import emap_oracle as eo  # noqa: E402
from oracle.class_max import ClassMaxOracle, decode_max  # noqa: E402
import _fixtures as fx  # noqa: E402


def _cloud(C, N, seed, ids):
    """x y z | two (probability, id) channels | one averaged feature"""
    from elevation_mapping_cupy_amd.fusion.pointcloud_class_max import encode_max
    rng = np.random.default_rng(seed)
    p = fx.cloud(C, N, seed)
    # the runner-up channel uses classes of its own: a class that is the maximum in SOME cell has its whole plane zeroed before the
    # next layer (pointcloud_class_max.py:121), so with shared classes the second layer would come out empty
    k1 = rng.choice(ids, N); k2 = rng.choice(np.asarray(ids) + 100, N)
    pr1 = rng.uniform(0.5, 1.0, N); pr2 = rng.uniform(0.0, 0.04, N)       # (a dozen runner-up points of one cell stay below one winner)
    return np.column_stack([p, encode_max(pr1, k1), encode_max(pr2, k2), rng.uniform(0, 1, N).astype(np.float32)]).astype(np.float32)


def test_encode_decode_roundtrip():
    from elevation_mapping_cupy_amd.fusion.pointcloud_class_max import encode_max
    pr = np.array([0.0, 0.25, 0.9995, 1.0], np.float32); ids = np.array([0, 7, 300, 65535], np.uint32)
    ma, ind = decode_max(encode_max(pr, ids))
    assert np.array_equal(ind, ids) and np.array_equal(ma, pr.astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_class_max_frames_against_the_restatement(scatter, weights):
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = 98, 20000
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    par = parameter_from(cfg, C, "reference_fp16", weights)
    par.pointcloud_channel_fusions = {"top.*": "class_max", "default": "average"}
    hip = ElevationMap(par)
    hip.set_scatter_mode(scatter)
    orc = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    CH = ["x", "y", "z", "top1", "top2", "feat"]
    R, t = fx.POSES["rotated"]
    cm = ClassMaxOracle(C)
    sem = np.zeros((3, C, C), np.float32); ids = np.zeros((2, C, C), np.uint32)
    class_sets = [[0, 1, 2, 3], [1, 2, 3, 4, 5], [0, 2, 9]]            # dense ids: the domain where the reference is defined
    for f, cls in enumerate(class_sets):
        p = _cloud(C, N, 30 + f, cls)
        hip.input_pointcloud(p, CH, R, t.copy() + hip.center, 0.0, 0.0)
        idx, valid, inside = orc.point_index(p, R, t)
        cm(p, idx, valid, inside, [3, 4], [0, 1], sem, ids)
        got = hip.semantic_map.semantic_map
        assert hip.semantic_map.layer_names[:2] == ["top1", "top2"]
        for k in range(2):
            assert np.array_equal(got[k].view(np.uint32), sem[k].view(np.uint32)), "frame %d layer %d: %d cells differ" % (
                f, k, (got[k] != sem[k]).sum())
            assert np.array_equal(hip.semantic_map.get_id_max(k), ids[k]), "frame %d id plane %d" % (f, k)
        plug = hip.semantic_map.fusion_manager.get_plugin("class_max", "pointcloud")
        assert np.array_equal(plug.unique_id, cm.unique_id)
        assert (got[2] != 0).sum() > 1000                               # the averaged channel of the same cloud was fused as well
    assert (sem[0] > 0).sum() > 2000 and (sem[1] > 0).sum() > 1000 and len(np.unique(ids[0])) >= 3
    # the normalised layers of a cell sum to one (or are all zero)
    tot = got[0] + got[1]
    assert np.all((np.abs(tot - 1) < 1e-6) | (tot == 0))


def test_id_planes_move_with_the_map(weights):
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = 66, 8000
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    par = parameter_from(cfg, C, "reference_fp16", weights)
    par.pointcloud_channel_fusions = {"top.*": "class_max"}
    hip = ElevationMap(par)
    R, t = fx.POSES["identity"]
    hip.input_pointcloud(_cloud(C, N, 3, [1, 2, 3])[:, :5], ["x", "y", "z", "top1", "top2"], R, t.copy() + hip.center, 0.0, 0.0)
    before = hip.semantic_map.get_id_max(0).copy()
    hip.move_to(np.array([0.04 * 5, -0.04 * 3, 0.0]), np.eye(3))        # 5 rows, 3 columns
    after = hip.semantic_map.get_id_max(0)
    moved = np.roll(before, (-5, 3), axis=(0, 1))
    inner = (slice(8, C - 8), slice(8, C - 8))
    assert np.array_equal(after[inner], moved[inner]) or np.array_equal(after[inner], np.roll(before, (5, -3), axis=(0, 1))[inner])
    assert (after != 0).sum() < (before != 0).sum()                     # the entering band reads 0
