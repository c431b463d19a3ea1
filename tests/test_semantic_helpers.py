"""CPU: the small host helpers of SemanticMap that the reference exposes next to the fusions (EM/semantic_map.py: pad_value :99-125,
get_layer_indices :184-197, decode_max :311-327, process_map_for_publish :376-386).  They are pure array code, so they are checked
without a device: against hand-made known answers, and -- only on EMAP_REF_EXEC=1, through the vetted AST path of oracle/ref_host.py --
against the reference's own statements executed from its file with NumPy standing in for CuPy."""
import ast

import numpy as np
import pytest

from elevation_mapping_cupy_amd.semantic_map import SemanticMap

SM = SemanticMap.__new__(SemanticMap)          # the helpers touch no state: no context, no device


def _packed(prob, cid):
    """(half probability | class id << 16) as the float32 bit pattern the class_max channels carry"""
    bits = np.asarray(prob, np.float16).view(np.uint16).astype(np.uint32) | (np.asarray(cid, np.uint32) << np.uint32(16))
    return bits.view(np.float32)


def test_decode_max_known_answers():
    prob = np.array([[0.0, 0.5, 1.0], [0.25, 0.999, 6.1e-5]], np.float16)
    cid = np.array([[0, 1, 65535], [7, 300, 2]], np.uint32)
    p, i = SM.decode_max(_packed(prob, cid))
    assert p.dtype == np.float32 and np.array_equal(p, prob.astype(np.float32))
    assert np.array_equal(i, cid)
    p, i = SM.decode_max(np.zeros((0, 2), np.float32))
    assert p.shape == (0, 2) and i.shape == (0, 2)


def test_get_layer_indices_keeps_the_reference_quirk():
    specs = {"rgb": "color", "class": "class_max", "class_b": "class_bayesian", "max": "class_max", "sem": "class_max"}
    # `key in val == fusion_alg` is a chained comparison: the layer's NAME must be a substring of its fusion's name
    assert SM.get_layer_indices("class_max", specs).tolist() == [1, 3]
    assert SM.get_layer_indices("class_bayesian", specs).tolist() == [2]         # "class_b" is a substring of "class_bayesian"; "sem" is not one of "class_max"
    assert SM.get_layer_indices("color", specs).tolist() == []
    assert SM.get_layer_indices("class_max", {}).dtype == np.int32


def test_pad_value_and_border_strip():
    x = np.arange(2 * 5 * 6, dtype=np.float32).reshape(2, 5, 6) + 1
    y = x.copy(); SM.pad_value(y, (2, -1))
    assert (y[:, :2] == 0).all() and (y[:, :, -1:] == 0).all() and np.array_equal(y[:, 2:, :-1], x[:, 2:, :-1])
    y = x.copy(); SM.pad_value(y, (-3, 4), idx=1, value=7.0)
    assert np.array_equal(y[0], x[0]) and (y[1, -3:] == 7).all() and (y[1, :, :4] == 7).all() and np.array_equal(y[1, :-3, 4:], x[1, :-3, 4:])
    y = x.copy(); SM.pad_value(y, (0, 0))
    assert np.array_equal(y, x)
    m = SM.process_map_for_publish(x[0])
    assert m.shape == (3, 4) and np.array_equal(m, x[0, 1:-1, 1:-1])
    m[:] = -1
    assert x[0, 1, 1] != -1                                                          # a copy, as in the reference


def test_helpers_against_the_reference_statements():
    """opt-in (EMAP_REF_EXEC=1): the four methods taken from the reference's semantic_map.py, vetted, compiled with cp = numpy"""
    import os
    from oracle import ref_host
    path = os.path.join(os.path.dirname(ref_host.REF_FILE), "semantic_map.py")
    if not (os.environ.get("EMAP_REF_EXEC", "0") == "1" and os.path.isfile(path)):
        pytest.skip("live execution of reference host code is opt-in (EMAP_REF_EXEC=1)")
    tree = ast.parse(open(path).read(), path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SemanticMap")
    names = ("pad_value", "get_layer_indices", "decode_max", "process_map_for_publish")
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names}
    ns = ref_host._namespace()
    ns["__builtins__"] = ref_host.exec_builtins(enumerate=enumerate)
    for n in names:
        ref_host._vet(fns[n], n, extra_builtins=("enumerate",))
    exec(compile(ast.Module(body=[fns[n] for n in names], type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(3)
    mer = _packed(rng.uniform(0, 1, (50, 3)).astype(np.float16), rng.integers(0, 60000, (50, 3)))
    for a, b in zip(SM.decode_max(mer), ns["decode_max"](None, mer)):
        assert np.array_equal(a, np.asarray(b))
    specs = {"rgb": "color", "class": "class_max", "class_b": "class_bayesian", "max": "class_max", "sem": "class_max", "class_bayesian": "class_bayesian"}
    for alg in ("class_max", "class_bayesian", "color", "average"):
        assert SM.get_layer_indices(alg, specs).tolist() == np.asarray(ns["get_layer_indices"](None, alg, specs)).tolist()
    x = rng.normal(size=(3, 9, 8)).astype(np.float32)
    for shift in ((2, -1), (-3, 4), (0, 0), (9, -8)):
        for idx in (None, 2):
            a, b = x.copy(), x.copy()
            SM.pad_value(a, shift, idx=idx, value=1.5); ns["pad_value"](None, b, shift, idx=idx, value=1.5)
            assert np.array_equal(a, b)
    assert np.array_equal(SM.process_map_for_publish(x[1]), ns["process_map_for_publish"](None, x[1]))


def test_elevation_map_pad_value_is_the_same_helper():
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    x = np.ones((7, 6, 6), np.float32)
    ElevationMap.pad_value(None, x, (1, -2), idx=1, value=9.0)
    assert (x[1, :1] == 9).all() and (x[1, :, -2:] == 9).all() and (x[0] == 1).all() and x[1, 1:, :-2].min() == 1
