"""CPU: provenance of tests/golden/.  The committed fixtures are what tests/golden/make_golden.py writes when it drives the REFERENCE'S
OWN kernel source compiled for the host (oracle/_ref: prebuilt objects travel with the repository; rebuilt from /root/reference where
that exists): regenerated here into a scratch directory, every array must come back bit for bit.  The two fixtures that EXECUTE
reference host code (host_steps.npz, class_max_ref66.npz: oracle/ref_host.py / ref_fusion.py) are regenerated only on
EMAP_REF_EXEC=1, like every other live execution of untrusted reference code."""
import importlib.util
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import build_ref, ref_kernels

G = os.path.join(ROOT, "tests", "golden")


def _generator(out_dir):
    spec = importlib.util.spec_from_file_location("make_golden_under_test", os.path.join(G, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    mk.OUT = str(out_dir)
    return mk


def _same(a, b):
    if a.endswith(".json"):
        return json.load(open(a)) == json.load(open(b))
    x, y = np.load(a), np.load(b)
    return set(x.files) == set(y.files) and all(x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and x[k].tobytes() == y[k].tobytes() for k in x.files)


def _need(*sets):
    for s in sets:
        if not ref_kernels.available(build_ref.PREBUILD[s]):
            pytest.skip("compiled reference object %s neither prebuilt nor buildable here" % s)


@pytest.mark.parametrize("step,sets,files", [
    ("semantic66", ("yaml66",), ("semantic_yaml66.npz",)),
    ("bayes66", ("bayes66",), ("bayes_yaml66.npz",)),
    ("semantic_toy", ("toy4",), ("semantic_toy.npz",)),
    ("warm_single", ("yaml202", "default202", "wall202"), ("warm_single.npz",)),
    ("frame66", ("yaml66",), ("frame_yaml66.npz",)),
    ("stencils", ("default34", "yaml66"), ("stencil.npz",)),
])
def test_committed_fixture_regenerates_bit_for_bit_from_the_compiled_reference(step, sets, files, tmp_path, capsys):
    _need(*sets)
    mk = _generator(tmp_path)
    getattr(mk, step)()
    capsys.readouterr()
    for f in files:
        assert os.path.exists(tmp_path / f), f
        assert _same(str(tmp_path / f), os.path.join(G, f)), "%s: the committed fixture is not what make_golden.py::%s writes" % (f, step)


@pytest.mark.parametrize("name,C,N", [("yaml202", 202, 50000), ("default202", 202, 50000)])
def test_known_answer_records_regenerate(name, C, N, tmp_path, capsys):
    _need(name)
    mk = _generator(tmp_path)
    mk.kat(name, build_ref.PREBUILD[name], C, N)
    capsys.readouterr()
    f = "kat_%s.json" % name
    assert _same(str(tmp_path / f), os.path.join(G, f))


@pytest.mark.parametrize("step,files", [("host_steps", ("host_steps.npz",)), ("class_max_ref", ("class_max_ref66.npz",))])
def test_fixtures_of_reference_host_code_regenerate(step, files, tmp_path, capsys):
    if os.environ.get("EMAP_REF_EXEC", "0") != "1" or not os.path.isdir(build_ref.REF_ROOT):
        pytest.skip("executes reference host code: opt-in (EMAP_REF_EXEC=1) and only where /root/reference exists")
    mk = _generator(tmp_path)
    getattr(mk, step)()
    capsys.readouterr()
    for f in files:
        assert _same(str(tmp_path / f), os.path.join(G, f)), f
