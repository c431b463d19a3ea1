"""TEST INFRASTRUCTURE ONLY.  Executes the reference's HOST code (array arithmetic of ``ElevationMap``) from the file where it lies
under /root/reference with NumPy standing in for CuPy.

``EM/elevation_mapping.py`` cannot be imported here (cupy, ruamel, simple_parsing, torch glue ...), but the host steps of the hot
path are pure array code: the drift gate (``update_map_with_kernel``, :346-357), ``clear_overlap_map`` (:393-410),
``update_variance`` / ``update_time`` (:420-426) and the map shift (``move_to`` / ``move`` / ``pad_value`` / ``shift_map_xy`` /
``shift_map_z``, :139-226).  This module parses the file with ``ast``, takes those function bodies (and the gate's ``if``
statement) unmodified, compiles them with ``cp = xp = numpy`` and binds them to a small state object.  Nothing of the reference
is copied into the repository.  The committed golden vectors (tests/golden/host_steps.npz, made by tests/golden/make_golden.py from
this module) are the DEFAULT pin; executing the reference file live is opt-in:

* ``available()`` is True only when ``EMAP_REF_EXEC=1`` is set AND the file exists (the reference tree is untrusted public content:
  nothing of it runs unless the person at the keyboard asks for it; ``EMAP_REF_FILE`` overrides the path for a hermetic regeneration);
* before anything is compiled, ``_vet`` walks the extracted syntax trees and rejects every construct the known host code does not
  need: imports, nested function / class definitions, lambdas, ``global`` / ``nonlocal``, ``with`` (except ``with self.map_lock``) / ``try`` / ``raise`` / ``del``,
  awaits / yields, dunder names and dunder attributes, string formatting, and every call or attribute chain whose root is not one of
  ``self`` / ``cp`` / ``xp`` / ``np`` / a local variable / a whitelisted builtin (``int``, ``float``, ``abs``, ``min``, ``max``, ``len``, ``range``,
  ``print``).  The namespace handed to ``exec`` carries no ``__builtins__`` beyond that list -- plus ``__import__``, which the vetted code
  cannot name (dunder names are rejected above) but NumPy's C code looks up in the CALLING frame's builtins the first time it lazily
  imports one of its own helper modules (``PyImport_Import``): without it the first array operation of a process that happens inside
  the extracted code dies with ``KeyError: '__import__'`` (seen when a live test ran on its own, round 5).
"""
from __future__ import annotations

import ast
import contextlib
import os
import types

import numpy as np

REF_FILE = os.environ.get("EMAP_REF_FILE", "/root/reference/elevation_mapping_cupy/script/elevation_mapping_cupy/elevation_mapping.py")
_SAFE_BUILTINS = {"int": int, "float": float, "abs": abs, "min": min, "max": max, "len": len, "range": range, "print": print, "bool": bool}
_ROOTS = {"self", "cp", "xp", "np"}
METHODS = ("clear_overlap_map", "update_variance", "update_time", "move", "move_to", "pad_value", "shift_map_xy", "shift_map_z",
           "shift_translation_to_map_center")


def exec_builtins(**extra):
    """builtins of the namespaces the vetted reference code runs in (see the module docstring for why ``__import__`` is there)"""
    return dict(_SAFE_BUILTINS, __import__=__import__, **extra)


def available():
    """live execution of the reference's host code: explicit opt-in only (the golden vectors are the default pin)"""
    return os.environ.get("EMAP_REF_EXEC", "0") == "1" and os.path.isfile(REF_FILE)


class UnsafeReferenceCode(RuntimeError):
    pass


def _vet(node, what, extra_locals=(), extra_builtins=()):
    """reject anything beyond plain array arithmetic on self / cp / xp / np before the tree is compiled (see the module docstring)"""
    banned = (ast.Import, ast.ImportFrom, ast.ClassDef, ast.Lambda, ast.Global, ast.Nonlocal, ast.AsyncWith, ast.Try, ast.Raise,
              ast.Delete, ast.Await, ast.Yield, ast.YieldFrom, ast.AsyncFunctionDef, ast.AsyncFor, ast.JoinedStr, ast.NamedExpr, ast.Starred)
    top = node
    local = set(extra_locals)
    for n in ast.walk(top):
        if isinstance(n, ast.arg):
            local.add(n.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            local.add(n.id)
    for n in ast.walk(top):
        if isinstance(n, banned) or (isinstance(n, ast.FunctionDef) and n is not top):
            raise UnsafeReferenceCode("%s: %s at line %s is not allowed" % (what, type(n).__name__, getattr(n, "lineno", "?")))
        if isinstance(n, ast.With):          # the reference guards its map with `with self.map_lock:` -- nothing else may be entered
            for it in n.items:
                e = it.context_expr
                if not (it.optional_vars is None and isinstance(e, ast.Attribute) and e.attr == "map_lock" and isinstance(e.value, ast.Name) and e.value.id == "self"):
                    raise UnsafeReferenceCode("%s: `with` on something other than self.map_lock (line %s)" % (what, n.lineno))
        if isinstance(n, ast.Name) and n.id.startswith("__"):
            raise UnsafeReferenceCode("%s: dunder name %r" % (what, n.id))
        if isinstance(n, ast.Attribute) and n.attr.startswith("_"):
            raise UnsafeReferenceCode("%s: private / dunder attribute %r" % (what, n.attr))
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in local and n.id not in _ROOTS and n.id not in _SAFE_BUILTINS and n.id not in extra_builtins:
            raise UnsafeReferenceCode("%s: free name %r at line %s" % (what, n.id, n.lineno))
        if isinstance(n, ast.Call):
            f = n.func
            while isinstance(f, (ast.Attribute, ast.Subscript, ast.Call)):
                f = f.value if not isinstance(f, ast.Call) else f.func
            if not (isinstance(f, ast.Name) and (f.id in _ROOTS or f.id in _SAFE_BUILTINS or f.id in extra_builtins or f.id in local)):
                raise UnsafeReferenceCode("%s: call rooted at %s (line %s)" % (what, ast.dump(f)[:60], n.lineno))


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
    return {"cp": xp, "xp": xp, "np": np, "__builtins__": exec_builtins()}


def load():
    """returns (class RefHost, gate_fn); RefHost carries the reference's methods named in METHODS"""
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ElevationMap")
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    ns = _namespace()
    for m in METHODS:
        _vet(fns[m], m)
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
    _vet(gate_if, "drift gate", ("self", "error", "error_cnt", "position_noise", "orientation_noise"))
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
