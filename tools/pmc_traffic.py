#!/usr/bin/env python3
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
    python tools/pmc_traffic.py gpurun_out/pmc_kitti00_FETCH_SIZE gpurun_out/pmc_kitti00_WRITE_SIZE kitti00 > profiles/r02_pmc_traffic.json
Units and corrections as in MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x, so the raw and the x2 sums are
both given (gather patterns are uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)  # newest run
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        n = n.replace("void ", "").split("<")[0]  # k_foo<double> / k_foo<float> -> k_foo
        acc[n].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    shape = sys.argv[3] if len(sys.argv) > 3 else "kitti00"
    out = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on: "
                  "python bench.py --workload %s --steps 1 --warmup 1 --no-cpu-baseline --no-extras, round 3 "
                  "(tools/refresh_profiles.sh)" % shape,
        "note": "FETCH_SIZE on gfx950 under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); "
                "both the raw and the x2-corrected sums are given; gather/scatter patterns are uncalibrated",
        "kernels": {},
    }
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {
            "FETCH_SIZE_KB_per_launch": fk,
            "WRITE_SIZE_KB_per_launch": wk,
            "hbm_bytes_per_launch_raw": 1024.0 * (fk + wk),
            "hbm_bytes_per_launch_fetch_x2": 1024.0 * (2 * fk + wk),
        }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
