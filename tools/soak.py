"""Repeated-use check (SLAM calls BA thousands of times): many initialize()/optimize() cycles on
graphs whose topology changes every cycle; device memory must stay flat and results reproducible.

    python tools/soak.py [cycles]          (needs an MI355X)
"""
import importlib
import os
import sys
import time

import ctypes

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        hip.hipDeviceSynchronize()
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    rng = np.random.default_rng(0)
    free0 = None
    chi_first = {}
    t0 = time.perf_counter()
    for c in range(cycles):
        kind = c % 4
        P, L, E, lc = [(30, 3000, 12600, 0), (160, 2500, 10500, 80), (200, 3000, 12500, 100),
                       (400, 8000, 33000, 200)][kind]
        d = cugo.synth(P, L, E, seed=kind + 1, n_loop_closures=lc, stereo_fraction=0.7)
        g = cugo.graph_from_arrays(d)
        if c % 3 == 0:
            g.set_float32(True)
        g.initialize()
        g.optimize(3)
        chi = tuple(s["chi2"] for s in g.stats())
        key = (kind, c % 3 == 0)
        if key in chi_first:
            assert chi == chi_first[key], (c, key, chi, chi_first[key])  # bitwise reproducible across cycles
        chi_first[key] = chi
        # second call on the same optimiser: structure reuse path
        g.initialize()
        g.optimize(2)
        g.close()
        if c == 8:
            free0 = free_bytes()
    free1 = free_bytes()
    print("cycles %d in %.1f s; device memory free after warm-up %.1f MiB, at the end %.1f MiB (drift %.1f MiB)" %
          (cycles, time.perf_counter() - t0, free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
    assert free0 - free1 < 64 * 2**20, "device memory keeps growing"
    print("soak ok")


if __name__ == "__main__":
    main()
