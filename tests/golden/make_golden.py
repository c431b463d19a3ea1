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
"""
import hashlib
import json
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import build_ref, ref_kernels  # noqa: E402
import _fixtures as fx  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def weights():
    w = pickle.load(open("/root/reference/elevation_mapping_cupy/config/core/weights.dat", "rb"))
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


if __name__ == "__main__":
    weights()
    kat("yaml202", build_ref.PREBUILD["yaml202"], 202, 50000)
    kat("default202", build_ref.PREBUILD["default202"], 202, 50000)
    kat("yaml1024", build_ref.PREBUILD["yaml1024"], 1024, 200000)
    frame66()
    stencils()
    print(sorted(os.listdir(OUT)))


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


if __name__ == "__main__":
    semantic66()
    bayes66()
    print(sorted(os.listdir(OUT)))
