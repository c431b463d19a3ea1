"""TEST INFRASTRUCTURE ONLY.  Recipe that compiles the *reference's own kernel source* for the host CPU.

The reference (leggedrobotics/elevation_mapping_cupy) keeps its whole numeric core as CUDA-C strings handed
to ``cupy.ElementwiseKernel`` (``elevation_mapping_cupy/script/elevation_mapping_cupy/kernels/custom_kernels.py``
and ``custom_semantic_kernels.py``).  CuPy and CUDA are absent here, so this script

1. installs a fake ``cupy`` module whose ``ElementwiseKernel`` merely records the strings,
2. imports the two reference kernel files *where they lie* under ``/root/reference`` and calls their
   factories with the requested parameter set (parameters are baked into the source as literals, exactly
   as the reference does, so one parameter set == one shared object),
3. wraps every captured ``operation`` in a sequential ``for (ptrdiff_t i = 0; i < size; ++i)`` loop (the body runs in a
   per-element lambda so that ``return;`` ends the element, also from inner loops) behind ``oracle/ref_shim.h`` and
4. builds ``oracle/_ref/ref_<hash>.so`` with ``g++ -O2 -mf16c -ffp-contract=off``.

Nothing from the reference is copied into the repository: the generated translation unit and the .so live
only under ``oracle/_ref/`` which is git-ignored (it still travels to the GPU box with ``gpurun``).
On a machine without ``/root/reference`` only already-built objects can be used (``load(..., build=False)``).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import re
import subprocess
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF_ROOT = "/root/reference/elevation_mapping_cupy/script/elevation_mapping_cupy"

# Parameter sets used by tests / golden generation.  Keys follow the reference's Parameter dataclass
# (elevation_mapping_cupy/script/elevation_mapping_cupy/parameter.py:137-216).
PARAM_DEFAULT = dict(  # dataclass defaults
    resolution=0.04, cell_n=202, sensor_noise_factor=0.05, mahalanobis_thresh=2.0, outlier_variance=0.01,
    drift_compensation_variance_inlier=0.1, traversability_inlier=0.1, wall_num_thresh=100,
    max_ray_length=2.0, cleanup_step=0.01, cleanup_cos_thresh=0.5, min_valid_distance=0.3,
    max_height_range=1.0, ramped_height_range_a=0.3, ramped_height_range_b=1.0, ramped_height_range_c=0.2,
    enable_edge_sharpen=True, enable_visibility_cleanup=True, max_variance=1.0, initial_variance=10.0,
    dilation_size=2, average_weight=0.5,
)
PARAM_YAML = dict(  # elevation_mapping_cupy/config/core/core_param.yaml
    PARAM_DEFAULT, drift_compensation_variance_inlier=0.1, traversability_inlier=0.9, wall_num_thresh=20,
    max_ray_length=10.0, cleanup_step=0.1, cleanup_cos_thresh=0.1, min_valid_distance=0.5,
    max_variance=100.0, initial_variance=1000.0, dilation_size=3,
)


def with_(base, **kw):
    d = dict(base)
    d.update(kw)
    return d


class _Captured:
    def __init__(self, in_params, out_params, operation, name="kernel", preamble="", **_):
        self.in_params, self.out_params = in_params, out_params
        self.operation, self.name, self.preamble = operation, name, preamble


def _load_reference_factories():
    fake = types.ModuleType("cupy")
    fake.ElementwiseKernel = _Captured
    saved = sys.modules.get("cupy")
    sys.modules["cupy"] = fake
    mods = {}
    try:
        for fn in ("custom_kernels", "custom_semantic_kernels", "custom_image_kernels"):
            spec = importlib.util.spec_from_file_location("_ref_" + fn, os.path.join(REF_ROOT, "kernels", fn + ".py"))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            mods[fn] = m
    finally:
        if saved is None:
            del sys.modules["cupy"]
        else:
            sys.modules["cupy"] = saved
    return mods


def _instantiate(p):
    """Call the reference factories the way ElevationMap.compile_kernels does
    (reference elevation_mapping.py:228-282) and the fusion plugins do."""
    m = _load_reference_factories()
    ck, sk, ik = m["custom_kernels"], m["custom_semantic_kernels"], m["custom_image_kernels"]
    C, res = p["cell_n"], p["resolution"]
    f32 = dict(U="float", T="float", W="int", V="float", B="bool")
    u32 = dict(f32, V="unsigned int")
    ks = {
        "add_points": (ck.add_points_kernel(
            res, C, C, p["sensor_noise_factor"], p["mahalanobis_thresh"], p["outlier_variance"],
            p["wall_num_thresh"], p["max_ray_length"], p["cleanup_step"], p["min_valid_distance"],
            p["max_height_range"], p["cleanup_cos_thresh"], p["ramped_height_range_a"],
            p["ramped_height_range_b"], p["ramped_height_range_c"], p["enable_edge_sharpen"],
            p["enable_visibility_cleanup"]), f32),
        "error_counting": (ck.error_counting_kernel(
            res, C, C, p["sensor_noise_factor"], p["mahalanobis_thresh"],
            p["drift_compensation_variance_inlier"], p["traversability_inlier"], p["min_valid_distance"],
            p["max_height_range"], p["ramped_height_range_a"], p["ramped_height_range_b"],
            p["ramped_height_range_c"]), f32),
        "average_map": (ck.average_map_kernel(C, C, p["max_variance"], p["initial_variance"]), f32),
        "dilation_filter": (ck.dilation_filter_kernel(C, C, p["dilation_size"]), f32),
        "normal_filter": (ck.normal_filter_kernel(C, C, res), f32),
        "sem_sum": (sk.sum_kernel(res, C, C), f32),
        "sem_average": (sk.average_kernel(C, C), f32),
        "sem_class_average": (sk.class_average_kernel(C, C, p["average_weight"]), f32),
        "sem_add_color": (sk.add_color_kernel(C, C), u32),
        "sem_color_average": (sk.color_average_kernel(C, C), u32),
    }
    if p.get("image_kernels"):
        # camera path (reference elevation_mapping.py:295-305, fusion/image_exponential.py:49-52, image_color.py):
        # x1, y1, z1, image sizes and map_idx are passed BY VALUE in the reference's calls (:540-554)
        ks["image_correspondence"] = (ik.image_to_map_correspondence_kernel(res, C, C, 0.10), f32,
                                      {"x1", "y1", "z1", "image_height", "image_width"})
        ks["image_exponential"] = (ik.exponential_correspondences_to_map_kernel(C, C, 0.7), f32, {"map_idx", "image_height", "image_width"})
        ks["image_color"] = (ik.color_correspondences_to_map_kernel(C, C), f32, {"map_idx", "image_height", "image_width"})
        ks["image_average"] = (ik.average_correspondences_to_map_kernel(C, C), f32, {"map_idx", "image_height", "image_width"})   # defined, never launched by the reference
    if p.get("bayes_kernels"):
        # point fusions that keep their kernels inside the plugin module (reference fusion/pointcloud_class_bayesian.py:12-53,
        # fusion/pointcloud_bayesian_inference.py:12-83)
        cb, bi = _fusion_module("pointcloud_class_bayesian"), _fusion_module("pointcloud_bayesian_inference")
        ks["alpha"] = (cb.alpha_kernel(res, C, C), f32)
        ks["sum_compact"] = (bi.sum_compact_kernel(res, C, C), f32)
        ks["bayesian_inference"] = (bi.bayesian_inference_kernel(C, C), f32)
    if p.get("sum_max_kernel"):          # EM/kernels/custom_semantic_kernels.py:89-123 (max_id is an int array in its caller, fusion/pointcloud_class_max.py)
        ks["sem_sum_max"] = (sk.sum_max_kernel(res, C, C), dict(f32, T="int"))
    if p.get("class_max_kernel"):        # the kernel the class_max fusion keeps in its own module (EM/fusion/pointcloud_class_max.py:12-47); max_id = the uint32 class positions
        ks["cmax_sum_max"] = (_fusion_module("pointcloud_class_max").sum_max_kernel(res, C, C), dict(f32, T="unsigned int"))
    if p.get("polygon_kernel"):
        # safety-polygon service (reference elevation_mapping.py:283, 837-889): `raw int16 polygon_n` is a concrete type
        ks["polygon_mask"] = (ck.polygon_mask_kernel(C, C, res), dict(f32, int16="short"))
    for extra in p.get("extra_dilation_sizes", ()):
        ks["dilation_filter_%d" % extra] = (ck.dilation_filter_kernel(C, C, extra), f32)
    for d in p.get("min_filter_sizes", ()):
        ks["min_filter_%d" % d] = (_min_filter_kernel(C, d), f32)
    for d in p.get("max_filter_sizes", ()):
        ks["max_filter_%d" % d] = (_min_filter_kernel(C, d, "max_filter", "MaxFilter", "max_filter_kernel"), f32)
    return ks


def _min_filter_kernel(C, d, module="min_filter", cls="MinFilter", attr="min_filter_kernel"):
    """MinFilter builds its kernel inside the plugin class (reference plugins/min_filter.py:22-82): instantiate the
    class under fake ``cupy`` / package modules and take the captured kernel."""
    fake = types.ModuleType("cupy")
    fake.ElementwiseKernel = _Captured
    fake.zeros = lambda *a, **k: None
    fake.ndarray = object
    pkg = types.ModuleType("_refplug")
    pkg.__path__ = []
    pm = types.ModuleType("_refplug.plugin_manager")

    class PluginBase:  # noqa: D401 - stand-in for the reference's ABC
        def __init__(self, *a, **k):
            pass
    pm.PluginBase = PluginBase
    saved = {k: sys.modules.get(k) for k in ("cupy", "_refplug", "_refplug.plugin_manager")}
    sys.modules.update({"cupy": fake, "_refplug": pkg, "_refplug.plugin_manager": pm})
    try:
        spec = importlib.util.spec_from_file_location("_refplug." + module, os.path.join(REF_ROOT, "plugins", module + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return getattr(getattr(m, cls)(cell_n=C, dilation_size=d, iteration_n=1), attr)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _fusion_module(name):
    """import reference fusion/<name>.py with fake ``cupy`` and a stand-in for its relative ``.fusion_manager`` import"""
    fake = types.ModuleType("cupy")
    fake.ElementwiseKernel = _Captured
    pkg = types.ModuleType("_reffus")
    pkg.__path__ = []
    fm = types.ModuleType("_reffus.fusion_manager")

    class FusionBase:
        pass
    fm.FusionBase = FusionBase
    saved = {k: sys.modules.get(k) for k in ("cupy", "_reffus", "_reffus.fusion_manager")}
    sys.modules.update({"cupy": fake, "_reffus": pkg, "_reffus.fusion_manager": fm})
    try:
        spec = importlib.util.spec_from_file_location("_reffus." + name, os.path.join(REF_ROOT, "fusion", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _parse_params(s):
    out = []
    for part in s.split(","):
        toks = part.split()
        if not toks:
            continue
        assert toks[0] == "raw", part
        out.append((toks[1], toks[2]))
    return out


def _emit(ks):
    src = ['#include "ref_shim.h"\n']
    for sym, spec in ks.items():
        k, types_ = spec[0], spec[1]
        scalars = spec[2] if len(spec) > 2 else set()
        params = _parse_params(k.in_params) + _parse_params(k.out_params)
        op = k.operation                       # runs inside a per-element lambda, so `return;` keeps its meaning even in inner loops
        args = ", ".join(("%s %s" % (types_[t], n)) if n in scalars else ("%s* %s_" % (types_[t], n)) for t, n in params)
        binds = "".join("  Raw<%s> %s{%s_};\n" % (types_[t], n, n) for t, n in params if n not in scalars)
        tds = "".join("typedef %s %s;\n" % (v, kk) for kk, v in types_.items())
        src.append(
            "namespace ns_%s {\n%s%s\nextern \"C\" void ref_%s(%s, long size_) {\n%s"
            "  for (ptrdiff_t i = 0; i < size_; ++i) {\n  [&]() {\n%s\n  }();\n  }\n}\n}\n" % (sym, tds, k.preamble, sym, args, binds, op))
    return "\n".join(src)


def key_of(p):
    return hashlib.sha1(json.dumps(p, sort_keys=True, default=str).encode()).hexdigest()[:12]


def so_path(p):
    return os.path.join(OUT, "ref_%s.so" % key_of(p))


def build(p, force=False):
    """Build (or reuse) the compiled-reference object for parameter set ``p``; returns its path."""
    path = so_path(p)
    if os.path.exists(path) and not force:
        return path
    if not os.path.isdir(REF_ROOT):
        raise FileNotFoundError("reference sources not present and %s not prebuilt" % path)
    os.makedirs(OUT, exist_ok=True)
    cpp = path[:-3] + ".cpp"
    with open(cpp, "w") as f:
        f.write(_emit(_instantiate(p)))
    with open(path[:-3] + ".json", "w") as f:
        json.dump(p, f, sort_keys=True, default=str)
    subprocess.check_call(["g++", "-O2", "-mf16c", "-ffp-contract=off", "-fno-fast-math", "-w", "-std=c++17",
                           "-shared", "-fPIC", "-I", HERE, cpp, "-o", path])
    return path


# parameter sets that are always prebuilt (they travel to the GPU box as .so files)
PREBUILD = {
    "default202": PARAM_DEFAULT,
    "yaml202": PARAM_YAML,
    "yaml1024": with_(PARAM_YAML, cell_n=1024),
    "yaml202_norays": with_(PARAM_YAML, enable_visibility_cleanup=False),
    "yaml1024_norays": with_(PARAM_YAML, cell_n=1024, enable_visibility_cleanup=False),
    "default34": with_(PARAM_DEFAULT, cell_n=34, extra_dilation_sizes=(1, 3, 10), min_filter_sizes=(1, 2)),
    "yaml66": with_(PARAM_YAML, cell_n=66, extra_dilation_sizes=(1, 2, 10)),
    "image98": with_(PARAM_YAML, cell_n=98, image_kernels=True),
    "bayes66": with_(PARAM_YAML, cell_n=66, bayes_kernels=True),
    "polygon130": with_(PARAM_DEFAULT, cell_n=130, polygon_kernel=True),
    "maxfilter34": with_(PARAM_DEFAULT, cell_n=34, max_filter_sizes=(1, 2)),
    # the toy maps of the reference's own kernel tests (EM/tests/test_semantic_kernels.py: 4 x 4 cells, resolution 0.9)
    "toy4": with_(PARAM_YAML, cell_n=4, resolution=0.9, bayes_kernels=True, sum_max_kernel=True),
    "classmax66": with_(PARAM_YAML, cell_n=66, class_max_kernel=True),
    # wall-skip fixture (tests/_warm.py): two drift inliers in a cell already exceed wall_num_thresh
    "wall202": with_(PARAM_DEFAULT, wall_num_thresh=1),
}


def build_all():
    if not os.path.isdir(REF_ROOT):
        return []
    return [build(p) for p in PREBUILD.values()]


if __name__ == "__main__":
    for pth in build_all():
        print(pth)
