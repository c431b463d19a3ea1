"""CPU: exhaustive / dense checks of the arithmetic shortcuts the HIP kernels take where a result feeds a DECISION.

  * sqrt_rn_ge1 (k_post normals): every float x >= 1, every 1-ulp start value -> correctly rounded (3.2e9 cases, ~10 core-seconds);
  * exp_neg_det (traversability epilogue, shared literally by oracle and kernel): within 3 ulp of the correctly rounded exp(-a)."""
import ctypes as ct
import os
import subprocess

import numpy as np

from oracle import emap_oracle as eo

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sqrt_correction_is_correctly_rounded_for_every_x_ge_1(tmp_path):
    exe = str(tmp_path / "sqrt_proof")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", os.path.join(HERE, "proofs", "sqrt_rn_ge1.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    n, bad = (int(tok.split("=")[1]) for tok in out.stdout.split())
    assert n == 3 * (0x7f800000 - 0x3f800000 + 1) and bad == 0


def test_deterministic_exp_is_within_3_ulp_of_exp():
    f = eo.lib().eo_exp_neg
    f.restype = ct.c_float; f.argtypes = [ct.c_float]
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.uniform(0, 3, 60000), rng.uniform(0, 90, 30000), 10.0 ** rng.uniform(-8, 0, 10000), [0.0, 87.0, 88.5, 103.0]]).astype(np.float32)
    got = np.array([f(float(x)) for x in a], np.float32)
    want64 = np.exp(-a.astype(np.float64))
    want = want64.astype(np.float32)
    normal = want > 1.2e-38
    ulp = np.spacing(np.abs(want[normal]))
    assert np.max(np.abs(got[normal].astype(np.float64) - want64[normal]) / ulp) <= 3.0
    assert np.all(np.abs(got[~normal].astype(np.float64) - want64[~normal]) <= 3 * 1.4e-45)
    assert f(200.0) == 0.0 and f(1e30) == 0.0 and f(float("inf")) == 0.0 and np.isnan(f(float("nan"))) and f(0.0) == 1.0
