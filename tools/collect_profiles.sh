#!/bin/bash
# after `gpurun -- bash tools/refresh_profiles.sh`: copy the newest summaries from gpurun_out/ into profiles/
set -e
cd "$(dirname "$0")/.."
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/r01_pmc_traffic.json
python tests/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/r01_pmc_summary.txt
python tests/prof_summary.py gpurun_out/prof_r01 timeline > profiles/r01_kernel_stats_summary.txt
cp "$(ls -t gpurun_out/prof_r01/runc/*_kernel_stats.csv | head -1)" profiles/r01_kernel_stats.csv
python - <<'PY'
import json
for src, dst in (("gpurun_out/bench_kitti00.json", "profiles/r01_bench_kitti00.json"),
                 ("gpurun_out/bench_synth10k.json", "profiles/r01_bench_synth10k_1gpu.json"),
                 ("gpurun_out/bench_kitti00_float32.json", "profiles/r01_bench_kitti00_float32.json")):
    line = [l for l in open(src) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(dst, "w"), indent=1)
d = json.load(open("profiles/r01_bench_kitti00.json"))
print({k: d[k] for k in ("value", "ms_per_step", "init_ms", "ba_10iter_seconds_incl_initialize")},
      d["parity"]["max_rel_chi2_diff_vs_cpu"], d["cpu_baseline"]["seconds"])
for k, v in d["kernels"].items():
    if "frac" in v:
        print("%-24s %8.1f us  %9.3f %s  frac %.4f" % (k, v["avg_ms"] * 1e3, v["achieved"], v["unit"], v["frac"]))
print("synth10k ms_per_step", json.load(open("profiles/r01_bench_synth10k_1gpu.json"))["ms_per_step"])
PY
head -12 profiles/r01_kernel_stats_summary.txt
