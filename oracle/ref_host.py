"""TEST INFRASTRUCTURE ONLY.  Executes the reference's HOST code (array arithmetic of ``ElevationMap``) from the file where it lies
under /root/reference with NumPy standing in for CuPy.

``EM/elevation_mapping.py`` cannot be imported here (cupy, ruamel, simple_parsing, torch glue ...), but the host steps of the hot
path are pure array code: the drift gate (``update_map_with_kernel``, :346-357), ``clear_overlap_map`` (:393-410),
``update_variance`` / ``update_time`` (:420-426) and the map shift (``move_to`` / ``move`` / ``pad_value`` / ``shift_map_xy`` /
``shift_map_z``, :139-226).  This module parses the file with ``ast``, takes those function bodies (and the gate's ``if``
statement) unmodified, compiles them with ``cp = xp = numpy`` and binds them to a small state object.  Nothing of the reference
is copied into the repository; without /root/reference ``available()`` is False and the committed golden vectors
(tests/golden/host_steps.npz, made by tests/golden/make_golden.py from this module) stand in.
"""
from __future__ import annotations

import ast
import contextlib
import os
import types

import numpy as np

REF_FILE = "/root/reference/elevation_mapping_cupy/script/elevation_mapping_cupy/elevation_mapping.py"
METHODS = ("clear_overlap_map", "update_variance", "update_time", "move", "move_to", "pad_value", "shift_map_xy", "shift_map_z",
           "shift_translation_to_map_center")


def available():
    return os.path.isfile(REF_FILE)


class _Sem:
    """the only SemanticMap member the extracted code touches (shift_map_xy, elevation_mapping.py:214)"""

    def __init__(self):
        self.shifts = []

    def shift_map_xy(self, shift_value):
        self.shifts.append(np.array(shift_value).copy())


def _namespace():
    xp = types.ModuleType("numpy_as_cupy")
    xp.__dict__.update(np.__dict__)
    xp.asnumpy = lambda a: np.asarray(a)          # cupy-only helper used by get_position
    return {"cp": xp, "xp": xp, "np": np}


def load():
    """returns (class RefHost, gate_fn); RefHost carries the reference's methods named in METHODS"""
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ElevationMap")
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    ns = _namespace()
    mod = ast.Module(body=[fns[m] for m in METHODS], type_ignores=[])
    exec(compile(mod, REF_FILE, "exec"), ns)
    # the drift gate: the `if self.param.enable_drift_compensation and error_cnt > ...` statement of update_map_with_kernel
    upd = fns["update_map_with_kernel"]
    gate_if = None
    for node in ast.walk(upd):
        if isinstance(node, ast.If) and "enable_drift_compensation" in ast.dump(node.test):
            gate_if = node
            break
    assert gate_if is not None, "drift gate not found in the reference"
    args = ast.arguments(posonlyargs=[], args=[ast.arg(a) for a in ("self", "error", "error_cnt", "position_noise", "orientation_noise")],
                         kwonlyargs=[], kw_defaults=[], defaults=[])
    gfn = ast.FunctionDef(name="drift_gate", args=args, body=[gate_if], decorator_list=[], lineno=gate_if.lineno, col_offset=0)
    gmod = ast.fix_missing_locations(ast.Module(body=[gfn], type_ignores=[]))
    exec(compile(gmod, REF_FILE, "exec"), ns)

    class RefHost:
        """state the extracted methods use, laid out as reference ElevationMap.__init__ does (:58-94)"""

        def __init__(self, cfg, cell_n):
            self.param = types.SimpleNamespace(**cfg)
            self.resolution = cfg["resolution"]
            self.cell_n = int(cell_n)
            self.data_type = np.float32
            self.center = np.array([0, 0, 0], dtype=np.float32)
            self.base_rotation = np.eye(3, dtype=np.float32)
            self.map_lock = contextlib.nullcontext()
            self.elevation_map = np.zeros((7, self.cell_n, self.cell_n), dtype=np.float32)
            self.initial_variance = cfg["initial_variance"]
            self.elevation_map[1] += self.initial_variance
            self.elevation_map[3] += 1.0
            cell_range = int(cfg["overlap_clear_range_xy"] / self.resolution)
            cell_range = np.clip(cell_range, 0, self.cell_n)
            self.cell_min = self.cell_n // 2 - cell_range // 2
            self.cell_max = self.cell_n // 2 + cell_range // 2
            self.mean_error = 0.0
            self.additive_mean_error = 0.0
            self.semantic_map = _Sem()

    for m in METHODS:
        setattr(RefHost, m, ns[m])
    RefHost.drift_gate = ns["drift_gate"]
    return RefHost
