#!/usr/bin/env python3
"""Host-only: build the Hsc block pattern of a synthetic graph, run the symbolic analysis and
print the stage schedule (fronts per stage, pivot widths, boundary sizes, children).
    python tools/plan_dump.py [n_poses n_landmarks n_edges]"""
import ctypes as C
import importlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
from test_host import plan_arrays  # noqa: E402


def covis(npose, e_pose, e_lm):
    order = np.argsort(e_lm, kind="stable")
    ep, el = e_pose[order], e_lm[order]
    pairs = set()
    start = 0
    n = len(el)
    while start < n:
        end = start
        while end < n and el[end] == el[start]:
            end += 1
        ps = np.unique(ep[start:end])
        for i in range(len(ps)):
            for j in range(i, len(ps)):
                pairs.add((ps[i], ps[j]))
        start = end
    rows = [[] for _ in range(npose)]
    for a, b in pairs:
        rows[a].append(b)
    rowptr = np.zeros(npose + 1, np.int32)
    colind = []
    for r in range(npose):
        rows[r].sort()
        rowptr[r + 1] = rowptr[r] + len(rows[r])
        colind += rows[r]
    return rowptr, np.array(colind, np.int32)


def main():
    a = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [1322, 133383, 561116]
    d = cugo.synth(a[0], a[1], a[2], seed=0, stereo_fraction=0.7, n_loop_closures=4000)
    fixed = d["pose_fixed"].astype(bool)
    idx = -np.ones(a[0], np.int64)
    idx[~fixed] = np.arange((~fixed).sum())
    keep = ~fixed[d["e_pose"]] & ~d["lm_fixed"].astype(bool)[d["e_lm"]]
    rowptr, colind = covis(int((~fixed).sum()), idx[d["e_pose"][keep]], d["e_lm"][keep])
    lib = cugo.lib()
    s = C.c_void_p()
    assert lib.cugo_chol_create(None, C.byref(s)) == 0
    assert lib.cugo_chol_analyze(s, len(rowptr) - 1, rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                 colind.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    pl = plan_arrays(lib, s)
    stp, tp, tf = pl["stage_task_ptr"], pl["task_ptr"], pl["task_fronts"]
    ncb, nb, cp = pl["ncb"], pl["nb"], pl["child_ptr"]
    nz, fl, ns_, nst, fb = C.c_double(), C.c_double(), C.c_int(), C.c_int(), C.c_double()
    lib.cugo_chol_stats(s, C.byref(nz), C.byref(fl), C.byref(ns_), C.byref(nst), C.byref(fb))
    print("stages", len(stp) - 1, "fronts", len(ncb), "nnzL %.3g flops %.3g front bytes %.3g" % (nz.value, fl.value, fb.value))
    for st in range(len(stp) - 1):
        fr = [f for t in range(stp[st], stp[st + 1]) for f in tf[tp[t]:tp[t + 1]]]
        w = [int(ncb[f]) for f in fr]
        r = [int(nb[f] - ncb[f]) for f in fr]
        ch = [int(cp[f + 1] - cp[f]) for f in fr]
        if len(fr) <= 8:
            print("stage %2d: %3d fronts  ncb %s  boundary %s  children %s" % (st, len(fr), w, r, ch))
        else:
            print("stage %2d: %3d fronts  ncb max %d mean %.1f  boundary max %d mean %.1f" % (st, len(fr), max(w), np.mean(w), max(r), np.mean(r)))


if __name__ == "__main__":
    main()
