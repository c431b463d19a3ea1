"""Generates the golden fixtures of tests/golden/ from the REFERENCE'S OWN kernel source compiled for the host
(oracle/build_ref.py, needs /root/reference) -- run in the build container, commit the outputs.

    python tests/golden/make_golden.py

Fixtures (all inputs are regenerated from seeds by tests/_fixtures.py, only outputs are stored):
  weights.npz            traversability-filter weights read from the reference's config/core/weights.dat
  kat_<set>.json         known answers of one frame on a fresh map (error_counting -> add_points -> average_map ->
                         dilation -> normal): SHA-1 of the cell-index column, valid/inside counts, plane sums
                         (SURVEY.md appendix D records the same run for yaml202)
  frame_yaml66.npz       full output planes of that sequence on a 66x66 map (7 map planes, dilated plane, normals,
                         idx/valid/inside columns) for identity and rotated poses
  stencil_*.npz          dilation (radius 1,2,3,10 incl. flat-index wrap) and normal filter outputs on random planes
  semantic_yaml66.npz    sum/average/class_average/colour kernels
  semantic_toy.npz       all nine raw semantic kernels on the 4 x 4 toy map of the reference's own kernel tests
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import build_ref, ref_kernels  # noqa: E402
import _fixtures as fx  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def weights():
    from elevation_mapping_cupy_amd.parameter import _WeightsUnpickler      # numpy arrays in a dict only: the file is untrusted content
    w = _WeightsUnpickler(open("/root/reference/elevation_mapping_cupy/config/core/weights.dat", "rb")).load()
    np.savez(os.path.join(OUT, "weights.npz"), w1=w["conv1.weight"], w2=w["conv2.weight"], w3=w["conv3.weight"],
             w_out=w["conv_final.weight"])


def one_frame(rk, C, p, R, t, init_var):
    m = np.zeros((7, C, C), np.float32); m[1] = init_var; m[3] = 1
    nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
    err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
    pr = p.copy()
    rk.error_counting(m, pr, R, t, nm, err, cnt)
    rk.add_points(R, t, nrm, pr, m, nm)
    newmap_sums = [float(nm[k].astype(np.float64).sum()) for k in range(5)]
    rk.average_map(nm, m)
    dil = np.zeros((C, C), np.float32); dm = np.zeros((C, C), np.float32)
    rk.dilation_filter(m[5].copy(), (m[2] + m[6]).copy(), dil, dm)
    no = np.zeros((3, C, C), np.float32)
    rk.normal_filter(dil, m[2].copy(), no)
    return dict(map=m, dil=dil, dmask=dm, normal=no, tail=pr, err=float(err[0]), err_cnt=float(cnt[0]), newmap_sums=newmap_sums)


def kat(name, params, C, N):
    rk = ref_kernels.RefKernels(params)
    out = {}
    for pose_name, (R, t) in fx.POSES.items():
        r = one_frame(rk, C, fx.cloud(C, N, 0), R.ravel().copy(), t.copy(), params["initial_variance"])
        idx = r["tail"][:, 0].astype("<i4")
        out[pose_name] = dict(
            idx_sha1=hashlib.sha1(idx.tobytes()).hexdigest(), idx_sum=int(idx.astype(np.int64).sum()),
            n_valid=int(r["tail"][:, 1].sum()), n_inside=int(r["tail"][:, 2].sum()), newmap_sums=r["newmap_sums"],
            valid_cells=int((r["map"][2] > 0.5).sum()),
            plane_sums=[float(r["map"][k].astype(np.float64).sum()) for k in range(7)],
            dil_sum=float(r["dil"].astype(np.float64).sum()), dmask_sum=float(r["dmask"].sum()),
            normal_sums=[float(r["normal"][k].astype(np.float64).sum()) for k in range(3)])
    json.dump(dict(cell_n=C, n_points=N, seed=0, poses=out), open(os.path.join(OUT, "kat_%s.json" % name), "w"), indent=1)


def frame66():
    params = build_ref.PREBUILD["yaml66"]
    rk = ref_kernels.RefKernels(params)
    C, N = 66, 6000
    save = {}
    for pose_name, (R, t) in fx.POSES.items():
        r = one_frame(rk, C, fx.cloud(C, N, 0), R.ravel().copy(), t.copy(), params["initial_variance"])
        save[pose_name + "_map"] = r["map"]; save[pose_name + "_dil"] = r["dil"]; save[pose_name + "_normal"] = r["normal"]
        save[pose_name + "_idx"] = r["tail"][:, 0].astype(np.int32)
        save[pose_name + "_flags"] = (r["tail"][:, 1].astype(np.uint8) | (r["tail"][:, 2].astype(np.uint8) << 1))
    np.savez_compressed(os.path.join(OUT, "frame_yaml66.npz"), **save)


def stencils():
    save = {}
    for setname, C, sizes in (("default34", 34, (None, 1, 3, 10)), ("yaml66", 66, (None, 1, 2, 10))):
        rk = ref_kernels.RefKernels(build_ref.PREBUILD[setname])
        plane, mask = fx.stencil_inputs(C, 7)
        for s in sizes:
            d = build_ref.PREBUILD[setname]["dilation_size"] if s is None else s
            out = np.zeros((C, C), np.float32); om = np.zeros((C, C), np.float32)
            rk.dilation_filter(plane, mask, out, om, size=s)
            save["%s_dil%d" % (setname, d)] = out
            save["%s_dilmask%d" % (setname, d)] = om
        no = np.zeros((3, C, C), np.float32)
        rk.normal_filter(plane, (mask > 0.5).astype(np.float32), no)
        save["%s_normal" % setname] = no
    np.savez_compressed(os.path.join(OUT, "stencil.npz"), **save)


def core():
    weights()
    kat("yaml202", build_ref.PREBUILD["yaml202"], 202, 50000)
    kat("default202", build_ref.PREBUILD["default202"], 202, 50000)
    kat("yaml1024", build_ref.PREBUILD["yaml1024"], 1024, 200000)
    frame66()
    stencils()


def semantic66():
    """reference semantic kernels (custom_semantic_kernels.py) fed with the clobbered point buffer of add_points."""
    params = build_ref.PREBUILD["yaml66"]
    rk = ref_kernels.RefKernels(params)
    C, N, K = 66, 6000, 4          # columns: x y z | s0 s1 (average) | c0 (class_average) | rgb (color)
    R, t = fx.POSES["rotated"]; Rf = R.ravel().copy()
    p = fx.semantic_cloud(C, N, 5)
    m = np.zeros((7, C, C), np.float32); m[1] = params["initial_variance"]; m[3] = 1
    nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
    err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
    xyz = np.ascontiguousarray(p[:, :3])
    rk.error_counting(m, xyz, Rf, t, nm, err, cnt); rk.add_points(Rf, t, nrm, xyz, m, nm)
    pc = p.copy(); pc[:, :3] = xyz                                   # clobbered: idx, valid, inside (custom_kernels.py:260-262)
    sem = np.zeros((4, C, C), np.float32); sem[2] = fx.semantic_prev(C)   # layer 2 has previous values (EMA branch)
    newmap = np.zeros((4, C, C), np.float32)
    i32 = lambda *a: np.array(a, np.int32)
    rk.sem_sum(pc, Rf, t, i32(3, 4), i32(0, 1), i32(3 + K, 2), sem, newmap, N * 2)
    rk.sem_average(newmap, i32(3, 4), i32(0, 1), i32(3 + K, 2), nm, sem, C * C * 2)
    newmap2 = np.zeros((4, C, C), np.float32)
    rk.sem_sum(pc, Rf, t, i32(5), i32(2), i32(3 + K, 1), sem, newmap2, N)
    rk.sem_class_average(newmap2, i32(5), i32(2), i32(3 + K, 1), nm, sem, C * C)
    color_map = np.zeros((4, C, C), np.uint32)
    rk.sem_add_color(pc, Rf, t, i32(6), i32(3), i32(3 + K, 1), color_map, N)
    rk.sem_color_average(color_map, i32(6), i32(3), i32(3 + K, 1), sem, C * C)
    np.savez_compressed(os.path.join(OUT, "semantic_yaml66.npz"), sem=sem, cnt=nm[2])


def bayes66():
    """class_bayesian (alpha kernel + renormalisation, K = 2 channels incl. the launch-size quirk) and bayesian_inference
    (sum_compact + bayesian_inference kernels) as reference fusion/pointcloud_class_bayesian.py:56-75 and
    fusion/pointcloud_bayesian_inference.py:101-122 run them."""
    params = build_ref.PREBUILD["bayes66"]
    rk = ref_kernels.RefKernels(params)
    C, N, K = 66, 6000, 4
    R, t = fx.POSES["rotated"]; Rf = R.ravel().copy()
    p = fx.bayes_cloud(C, N, 5)
    m = np.zeros((7, C, C), np.float32); m[1] = params["initial_variance"]; m[3] = 1
    nm = np.zeros((7, C, C), np.float32); nrm = np.zeros((3, C, C), np.float32)
    err = np.zeros(1, np.float32); cnt = np.zeros(1, np.float32)
    xyz = np.ascontiguousarray(p[:, :3])
    rk.error_counting(m, xyz, Rf, t, nm, err, cnt); rk.add_points(Rf, t, nrm, xyz, m, nm)
    pc = p.copy(); pc[:, :3] = xyz
    i32 = lambda *a: np.array(a, np.int32)
    sem = np.zeros((3, C, C), np.float32); sem[2] = fx.semantic_prev(C)
    newmap = np.zeros((3, C, C), np.float32); newmap[:2] = fx.bayes_alpha_prior(C)       # persistent layers (semantic_map.py:54-56)
    rk.alpha(pc, i32(3, 4), i32(0, 1), i32(3 + K, 2), newmap, N)
    sum_alpha = np.sum(newmap[[0, 1]], axis=0); sum_alpha[sum_alpha == 0] = 1
    sem[[0, 1]] = newmap[[0, 1]] / np.expand_dims(sum_alpha, axis=0)
    sum_mean = np.zeros((1, C, C), np.float32)
    rk.sum_compact(pc, Rf, t, i32(5), i32(2), i32(3 + K, 1), sum_mean, N)
    rk.bayesian_inference(i32(5), i32(2), i32(3 + K, 1), nm, newmap, sum_mean, sem, C * C)
    np.savez_compressed(os.path.join(OUT, "bayes_yaml66.npz"), sem=sem, alpha=newmap[:2], sum_mean=sum_mean)




def warm_single():
    """race-free warm fixtures (tests/_warm.py): K one-point frames through the compiled reference on an injected warm map, for
    both parameter sets; stored as the sparse difference to the untouched map (+ totals that prove the branches were reached)."""
    import _warm as W
    from oracle import emap_oracle as eo
    save = {}
    for name, cfgname in W.SETS.items():
        cfg = getattr(eo, cfgname)
        rk = ref_kernels.RefKernels(build_ref.PREBUILD[name])
        C = 202
        m0, nrm = fx.warm_map(C, 1, cfg["initial_variance"])
        m = m0.copy()
        om = eo.OracleMap(eo.make_params(cfg, cell_n=C)); om.elevation_map[...] = m0; om.normal_map[...] = nrm
        tot = [0, 0]

        def frame(p, R, t):
            W.ref_frame(rk, m, nrm, p, R, t)
            h, o = W.oracle_frame(om, p, R, t); tot[0] += h; tot[1] += o

        def tick():
            m[4] += np.float32(cfg["time_interval"]); om.update_time()
        W.run_single(cfg, C, 1, frame, None, tick)
        idx, val = W.sparse_diff(W.base_single(m0, cfg), m)
        save[name + "_idx"], save[name + "_val"] = idx, val
        save[name + "_hits_outliers"] = np.array(tot, np.int64)      # counted by the contract oracle on the same frames
        print(name, "entries", idx.size, "ray hits", tot[0], "outliers", tot[1])
    # wall-skip fixture
    cfg = dict(eo.DEFAULTS, **W.WALL_CFG)
    rk = ref_kernels.RefKernels(build_ref.PREBUILD["wall202"])
    C = 202
    R, t = fx.POSES["identity"]
    m0, nrm = fx.warm_map(C, 2, cfg["initial_variance"])
    m, nrm = m0.copy(), nrm.copy()
    valid_after, order_dependent, want_skip = [], 0, []
    for ix, iy, cell, d3, pts, skipped in W.wall_sequence(C):
        m[:, ix, iy] = cell; nrm[:, ix, iy] = d3
        mr = m.copy(); W.ref_frame(rk, mr, nrm, pts[::-1].copy(), R, t)
        W.ref_frame(rk, m, nrm, pts, R, t)
        order_dependent += not all(np.allclose(m[q], mr[q], atol=1e-6, rtol=1e-6) for q in range(7))
        valid_after.append(m[2, ix, iy]); want_skip.append(skipped)
    assert order_dependent == 0, "wall fixture must be race free (forward == reversed point order)"
    idx, val = W.sparse_diff(W.base_after_reset(m0, np.float32(cfg["initial_variance"])), m)
    save["wall202_idx"], save["wall202_val"] = idx, val
    save["wall202_valid_after"] = np.array(valid_after, np.float32)
    assert all((v == 1.0) == sk for v, sk in zip(valid_after, want_skip)), "wall fixture: skip / penetration pattern not reached"
    print("wall202 entries", idx.size, "skipped", int((np.array(valid_after) == 1).sum()), "penetrated", int((np.array(valid_after) < 1).sum()))
    np.savez_compressed(os.path.join(OUT, "warm_single.npz"), **save)


def host_steps():
    """the reference's HOST code of the path executed with NumPy as cupy (oracle/ref_host.py): drift gate, overlap clearance,
    variance / time decay and a move_to / move sequence."""
    import _warm as W
    from oracle import emap_oracle as eo, ref_host
    H = ref_host.load()
    save = {}
    for cname in ("YAML", "DEFAULTS"):
        cfg = dict(getattr(eo, cname))
        # drift gate (elevation_mapping.py:346-357)
        rows = []
        for err, cnt, pn, on in GATE_CASES:
            h = H(cfg, 34); h.elevation_map[0] = fx.stencil_inputs(34, 3)[0]
            h.additive_mean_error = np.float32(0.25)
            h.drift_gate(np.array([err], np.float32), np.array([cnt], np.float32), pn, on)
            rows.append([float(np.asarray(h.mean_error).ravel()[0]), float(np.asarray(h.additive_mean_error).ravel()[0]), float(h.elevation_map[0, 5, 7])])
        save[cname + "_gate"] = np.array(rows, np.float64)
        # overlap clearance (:393-410), variance / time decay (:420-426)
        C = 130
        h = H(cfg, C); m0, _ = fx.warm_map(C, 3, cfg["initial_variance"]); h.elevation_map[...] = m0
        tz = np.array([0.1, -0.2, 2.6], np.float32)
        h.clear_overlap_map(tz)
        save[cname + "_overlap_idx"], save[cname + "_overlap_val"] = W.sparse_diff(m0, h.elevation_map)
        h.update_variance(); h.update_time()
        save[cname + "_decay_var_time"] = h.elevation_map[[1, 4]].copy()
    # shift sequence (:139-226)
    cfg = dict(eo.YAML); C = 34
    h = H(cfg, C); m0, _ = fx.warm_map(C, 4, cfg["initial_variance"]); h.elevation_map[...] = m0
    centers = []
    for kind, vec in MOVE_SEQUENCE:
        if kind == "move_to":
            h.move_to(np.array(vec, np.float64), np.eye(3))
        else:
            h.move(np.array(vec, np.float64))
        centers.append(np.asarray(h.center, np.float32).copy())
    save["shift_map"] = h.elevation_map.copy(); save["shift_centers"] = np.array(centers)
    save["shift_sem_calls"] = np.array(h.semantic_map.shifts, np.int32)
    np.savez_compressed(os.path.join(OUT, "host_steps.npz"), **save)
    print("host steps:", sorted(save))


def semantic_toy():
    """the nine raw semantic kernels (EM/kernels/custom_semantic_kernels.py) on the toy inputs of tests/_fixtures.py:
    semantic_kernel_cases, through the reference's own kernel source compiled for the host (parameter set toy4)"""
    rk = ref_kernels.RefKernels(build_ref.PREBUILD["toy4"])
    save = {}
    for case, c in fx.semantic_kernel_cases().items():
        for name, arr in fx.semantic_kernel_run(rk, c).items():
            save["%s_%s" % (case, name)] = arr
    np.savez_compressed(os.path.join(OUT, "semantic_toy.npz"), **save)
    print("semantic toy:", sorted(save))


def class_max_ref():
    """the class_max fusion as the REFERENCE computes it: ClassMax.decode_max / __call__ executed from the reference file (oracle/ref_fusion.py:
    NumPy with CuPy's gather semantics) over its own sum_max_kernel compiled for the host (parameter set classmax66), on the four
    frames of tests/_classmax.py; cell indices / flags of the points from the oracle (pinned elsewhere against the reference kernels)"""
    import _classmax as cmx
    from oracle import emap_oracle as eo, ref_fusion
    os.environ["EMAP_REF_EXEC"] = "1"                    # regeneration = the explicit request to execute the reference's host code
    rk = ref_kernels.RefKernels(build_ref.PREBUILD["classmax66"])
    RefClassMax = ref_fusion.load(rk)
    orc = eo.OracleMap(eo.make_params(dict(eo.YAML, enable_visibility_cleanup=False), cell_n=cmx.C))
    R, t = fx.POSES["rotated"]
    save = {}
    for f, (sem, ids, uniq) in enumerate(cmx.reference_frames(RefClassMax, lambda p: orc.point_index(p, R, t))):
        save["sem%d" % f] = sem; save["ids%d" % f] = ids; save["unique%d" % f] = uniq
        print("class_max frame", f, "unique_id", uniq.tolist(), "cells", int((sem[0] > 0).sum()), int((sem[1] > 0).sum()))
    np.savez_compressed(os.path.join(OUT, "class_max_ref66.npz"), **save)


GATE_CASES = [(12.5, 200, 1.0, 1.0), (-3.0, 150, 0.0, 1.0), (40.0, 200, 1.0, 0.0), (5.0, 100, 1.0, 1.0), (5.0, 101, 0.0, 0.0),
              (0.0, 0, 1.0, 1.0), (-19.0, 200, 1.0, 1.0)]
MOVE_SEQUENCE = [("move_to", (0.13, -0.3, 0.05)), ("move_to", (0.13, -0.3, 0.05)), ("move", (-0.21, 0.09, -0.02)),
                 ("move_to", (0.5, 0.5, 0.0)), ("move", (0.0, 0.0, 0.3)), ("move_to", (-0.9, 0.46, 0.11)), ("move", (2.0, -0.04, 0.0))]


if __name__ == "__main__":      # python tests/golden/make_golden.py [core semantic66 bayes66 warm_single host_steps]
    todo = sys.argv[1:] or ["core", "semantic66", "bayes66", "warm_single", "host_steps", "semantic_toy", "class_max_ref"]
    for name in todo:
        globals()[name]()
    print(sorted(os.listdir(OUT)))
