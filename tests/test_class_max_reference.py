"""The ``pointcloud_class_max`` fusion against the REFERENCE'S OWN host statements: ``ClassMax.decode_max`` / ``ClassMax.__call__``
(EM/fusion/pointcloud_class_max.py:62-126) executed from the reference file (oracle/ref_fusion.py: the syntax tree of the two
methods, vetted, NumPy with CuPy's gather semantics) over the fusion's own ``sum_max_kernel`` compiled for the host -- committed as
tests/golden/class_max_ref66.npz (tests/golden/make_golden.py: class_max_ref; regenerated live with EMAP_REF_EXEC=1).

CPU: the NumPy restatement (oracle/class_max.py) against it.  GPU: the product (``ElevationMap.input_pointcloud`` with a class_max
channel mapping) against it.  What must agree exactly: the id table (``unique_id``, including the frame whose stored ids index
beyond the table: CuPy wraps), the class-id planes, which cells carry a probability.  What may differ in the last bits: the
probabilities -- restatement and product sum the half-precision inputs EXACTLY and round once, the reference's float atomics round
after every addition in whatever order the GPU picks (the compiled kernel: point order) -- and, where two classes tie within that
error, the id the argmax picks."""
import os

import numpy as np
import pytest

import _classmax as cmx
import _fixtures as fx
from oracle import emap_oracle as eo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "class_max_ref66.npz")


def _compare(frames, gold, what):
    for f, (sem, ids, uniq) in enumerate(frames):
        assert np.array_equal(np.asarray(uniq, np.uint32), gold["unique%d" % f]), "%s frame %d: id table %s vs %s" % (what, f, list(uniq), gold["unique%d" % f].tolist())
        gs, gi = gold["sem%d" % f], gold["ids%d" % f]
        assert np.array_equal(sem > 0, gs > 0), "%s frame %d: different cells carry a probability" % (what, f)
        assert np.allclose(sem, gs, rtol=2e-6, atol=1e-7), "%s frame %d: max |d| = %g" % (what, f, np.abs(sem - gs).max())
        bad = ids != gi
        assert bad.sum() <= 2, "%s frame %d: %d class ids differ" % (what, f, int(bad.sum()))       # (near-ties of two classes' sums)
    assert len(frames) == len(cmx.CLASS_SETS)


def test_restatement_vs_reference_statements():
    from oracle.class_max import ClassMaxOracle
    gold = np.load(GOLD)
    orc = eo.OracleMap(eo.make_params(dict(eo.YAML, enable_visibility_cleanup=False), cell_n=cmx.C))
    R, t = fx.POSES["rotated"]
    cm = ClassMaxOracle(cmx.C)
    sem = np.zeros((2, cmx.C, cmx.C), np.float32); ids = np.zeros((2, cmx.C, cmx.C), np.uint32)
    frames = []
    for f in range(len(cmx.CLASS_SETS)):
        p = cmx.cloud(f)
        idx, valid, inside = orc.point_index(p, R, t)
        cm(p, idx, valid, inside, cmx.PCL_IDS, cmx.LAYER_IDS, sem, ids)
        frames.append((sem.copy(), ids.copy(), cm.unique_id.copy()))
    _compare(frames, gold, "restatement")
    assert gold["unique3"].tolist() == [0, 1, 2, 4, 7, 9, 102, 107, 300, 400]      # frame 4: the stored ids 100 .. 109 index a 12-entry table and wrap


def test_golden_regenerates_from_the_reference_file():
    """opt-in (EMAP_REF_EXEC=1, the reference tree present): execute the reference's statements again and compare with the committed file"""
    from oracle import build_ref, ref_fusion, ref_kernels
    if not ref_fusion.available() or not ref_kernels.available(build_ref.PREBUILD["classmax66"]):
        pytest.skip("live execution of the reference's host code is opt-in (EMAP_REF_EXEC=1) and needs /root/reference")
    gold = np.load(GOLD)
    RefClassMax = ref_fusion.load(ref_kernels.RefKernels(build_ref.PREBUILD["classmax66"]))
    orc = eo.OracleMap(eo.make_params(dict(eo.YAML, enable_visibility_cleanup=False), cell_n=cmx.C))
    R, t = fx.POSES["rotated"]
    for f, (sem, ids, uniq) in enumerate(cmx.reference_frames(RefClassMax, lambda p: orc.point_index(p, R, t))):
        assert np.array_equal(sem, gold["sem%d" % f]) and np.array_equal(ids, gold["ids%d" % f]) and np.array_equal(uniq, gold["unique%d" % f])


def test_vetting_refuses_code_beyond_array_arithmetic(tmp_path, monkeypatch):
    """the allow-list in front of the reference's fusion code: an import smuggled into __call__ is refused before anything runs"""
    from oracle import ref_fusion, ref_host
    src = "class ClassMax:\n    def decode_max(self, mer):\n        return mer, mer\n    def __call__(self, points_all):\n        import os\n        return os.getcwd()\n"
    f = tmp_path / "pointcloud_class_max.py"
    f.write_text(src)
    monkeypatch.setattr(ref_fusion, "REF_FILE", str(f))
    with pytest.raises(ref_host.UnsafeReferenceCode):
        ref_fusion.load(None)


@pytest.mark.gpu
@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_product_vs_reference_statements(scatter, weights):
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    gold = np.load(GOLD)
    par = parameter_from(dict(eo.YAML, enable_visibility_cleanup=False), cmx.C, "reference_fp16", weights)
    par.pointcloud_channel_fusions = {"top.*": "class_max"}
    hip = ElevationMap(par)
    hip.set_scatter_mode(scatter)
    R, t = fx.POSES["rotated"]
    frames = []
    for f in range(len(cmx.CLASS_SETS)):
        hip.input_pointcloud(cmx.cloud(f), cmx.CH, R, t.copy() + hip.center, 0.0, 0.0)
        plug = hip.semantic_map.fusion_manager.get_plugin("class_max", "pointcloud")
        frames.append((hip.semantic_map.semantic_map[:2].copy(), np.stack([hip.semantic_map.get_id_max(k) for k in range(2)]), np.asarray(plug.unique_id, np.uint32).copy()))
    _compare(frames, gold, "product (%s)" % scatter)
