"""Known-answer test against the ONLY result pins the reference holds for this path: the chi2 table
and the CPU-vs-GPU RMSE that its README prints for ba_kitti_00.json
(/root/reference/README.md:127-137, 162-178).

The dataset itself is not in the container (samples/ba_input.7z is a stripped large blob,
/root/reference/.MISSING_LARGE_BLOBS:1), so the two real legs skip until the file is supplied:

    CUGO_KITTI00_JSON=/path/to/ba_kitti_00.json python -m pytest tests/test_kat_kitti00.py

Everything else of the KAT — the loader of the reference's JSON layout, the sample's protocol
(ref samples/sample_ba_from_file/main.cpp:168-190: warm-up initialize(); optimize(1) which MUTATES
the estimates, then initialize(); optimize(10) whose ten chi2 values are the table), the
printed-to-0.1 comparison and the RMSE triplet — runs on every test run against a synthetic file in
the same format, so the dormant legs cannot rot.
"""
import importlib
import os

import numpy as np
import pytest

from conftest import ROOT

cugo = importlib.import_module("cuda-bundle-adjustment_amd")

# /root/reference/README.md:127-137 (sample_ba_from_file) == :162-173 (g2o CPU and GPU columns)
README_CHI2 = [334210.0, 331822.8, 329700.4, 327743.4, 326123.2, 324876.6, 323698.5, 322572.7, 321410.3,
               320086.4]
README_SIZES = (1322, 133383, 561116)          # README.md:107-110
# README.md:175-178: RMSE between the g2o (CPU) and the GPU estimates
README_RMSE = dict(rotation=7.63e-16, translation=4.50e-13, landmark=4.50e-13)


def kitti00_path():
    for p in (os.environ.get("CUGO_KITTI00_JSON", ""), os.path.join(ROOT, "samples", "ba_input", "ba_kitti_00.json"),
              os.path.join(ROOT, "tests", "golden", "ba_kitti_00.json")):
        if p and os.path.exists(p):
            return p
    return None


def matches_printed(chi, table):
    """the reference prints chi2 with %.1f: agreement to the printed digit"""
    return [abs(round(c, 1) - t) < 1e-6 or abs(c - t) <= 0.05 + 1e-9 for c, t in zip(chi, table)]


def sorted_by_id(d, pid, lid):
    """the loader's arrays are in FILE order; the oracle wants the vertices in ascending-id order
    (std::map iteration order, ref src/optimisable_graph.hpp:95)"""
    po, lo = np.argsort(pid), np.argsort(lid)
    inv_p, inv_l = np.empty_like(po), np.empty_like(lo)
    inv_p[po], inv_l[lo] = np.arange(len(po)), np.arange(len(lo))
    return dict(d, pose=d["pose"][po], pose_fixed=d["pose_fixed"][po], lm=d["lm"][lo], lm_fixed=d["lm_fixed"][lo],
                e_pose=inv_p[d["e_pose"]].astype(np.int32), e_lm=inv_l[d["e_lm"]].astype(np.int32))


def oracle_leg(oracle, d):
    """CPU restatement under the sample's protocol; returns (chi2 of the 10 counted iterations, problem)"""
    prob = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                          d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    assert len(prob.optimize(1)) == 1          # warm-up: one LM iteration, estimates updated in place
    return [r["chi2"] for r in prob.optimize(10)], prob


def gpu_leg(d, pose_ids, lm_ids):
    g = cugo.graph_from_arrays(d, pose_ids=pose_ids, lm_ids=lm_ids)
    g.initialize(); g.optimize(1)              # warm-up (ref main.cpp:168-169)
    g.initialize(); g.optimize(10)             # the timed / printed run (ref main.cpp:185-186)
    chi = [s["chi2"] for s in g.stats()]
    pose, lm = g.poses(pose_ids), g.landmarks(lm_ids)
    g.close()
    return chi, pose, lm


def rmse_triplet(pose_a, lm_a, pose_b, lm_b):
    """ref samples/sample_comparison_with_g2o/main.cpp:124-148: quaternion coefficients,
    translation, landmark position"""
    f = lambda a, b: float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b)) ** 2)))
    return dict(rotation=f(pose_a[:, :4], pose_b[:, :4]), translation=f(pose_a[:, 4:], pose_b[:, 4:]),
                landmark=f(lm_a, lm_b))


# --------------------------------------------------------------------------- real dataset ---
def test_readme_table_oracle_leg(oracle_lib):
    path = kitti00_path()
    if path is None:
        pytest.skip("ba_kitti_00.json not available (samples/ba_input.7z is stripped from the reference)")
    d, pid, lid = cugo.load_ba_json(path)
    assert (len(pid), len(lid), len(d["e_pose"])) == README_SIZES
    chi, _ = oracle_leg(oracle_lib, sorted_by_id(d, pid, lid))
    assert len(chi) == 10 and all(matches_printed(chi, README_CHI2)), list(zip(chi, README_CHI2))


@pytest.mark.gpu
def test_readme_table_gpu_leg(oracle_lib):
    path = kitti00_path()
    if path is None:
        pytest.skip("ba_kitti_00.json not available (samples/ba_input.7z is stripped from the reference)")
    d, pid, lid = cugo.load_ba_json(path)
    chi, pose, lm = gpu_leg(d, pid, lid)
    assert len(chi) == 10 and all(matches_printed(chi, README_CHI2)), list(zip(chi, README_CHI2))
    ref_chi, prob = oracle_leg(oracle_lib, sorted_by_id(d, pid, lid))
    for a, b in zip(chi, ref_chi):
        assert abs(a - b) <= 1e-10 * abs(b)
    r = rmse_triplet(pose[np.argsort(pid)], lm[np.argsort(lid)], prob.pose, prob.lm)
    # the README's own CPU-vs-GPU agreement, one decade of slack (another summation order)
    assert r["rotation"] < 10 * README_RMSE["rotation"] + 1e-15, r
    assert r["translation"] < 10 * README_RMSE["translation"] and r["landmark"] < 10 * README_RMSE["landmark"], r


# ------------------------------------------------------------------- the KAT's own machinery ---
def _synthetic_file(tmp_path, oracle):
    d = cugo.synth(40, 500, 2100, seed=12)
    pose_ids = np.arange(40) * 3 + 11
    lm_ids = np.arange(500)[::-1] + 5000       # descending: file order != id order
    path = str(tmp_path / "ba_synth.json")
    cugo.save_ba_json(path, d, pose_ids, lm_ids)
    # expected table by the flat-array path, without the loader or the protocol helper; the oracle
    # wants vertices in ascending-id order, so the landmarks are reversed for it
    prob = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"][::-1], d["lm_fixed"][::-1], d["e_pose"],
                          499 - d["e_lm"], d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    prob.optimize(1)
    table = [round(r["chi2"], 1) for r in prob.optimize(10)]
    return path, table, prob


def test_kat_machinery_on_synthetic_file_cpu(oracle_lib, tmp_path):
    path, table, _ = _synthetic_file(tmp_path, oracle_lib)
    d, pid, lid = cugo.load_ba_json(path)
    ds = sorted_by_id(d, pid, lid)
    chi, _ = oracle_leg(oracle_lib, ds)
    assert all(matches_printed(chi, table)), list(zip(chi, table))
    # the comparison really discriminates: the un-warmed-up trajectory does not match the table
    prob = oracle_lib.Problem(ds["pose"], ds["pose_fixed"], ds["lm"], ds["lm_fixed"], ds["e_pose"], ds["e_lm"],
                              ds["e_stereo"], ds["e_meas"], ds["e_omega"], ds["e_cam"])
    cold = [r["chi2"] for r in prob.optimize(10)]
    assert not all(matches_printed(cold, table))
    assert matches_printed([100.04, 100.06], [100.0, 100.0]) == [True, False]


@pytest.mark.gpu
def test_kat_machinery_on_synthetic_file_gpu(oracle_lib, tmp_path):
    path, table, prob = _synthetic_file(tmp_path, oracle_lib)
    d, pid, lid = cugo.load_ba_json(path)
    chi, pose, lm = gpu_leg(d, pid, lid)
    assert len(chi) == 10 and all(matches_printed(chi, table)), list(zip(chi, table))
    # estimates come back by id: oracle order = ascending id
    r = rmse_triplet(pose[np.argsort(pid)], lm[np.argsort(lid)], prob.pose, prob.lm)
    assert r["rotation"] < 1e-13 and r["translation"] < 1e-11 and r["landmark"] < 1e-10, r
