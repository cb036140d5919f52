#!/bin/bash
# after `gpurun -- bash tools/refresh_profiles.sh`: copy the newest summaries from gpurun_out/ into profiles/
set -e
R=${R:-r04}
cd "$(dirname "$0")/.."
for W in kitti00 synth10k; do
  sfx=""; [ $W = kitti00 ] || sfx="_$W"
  python tools/pmc_traffic.py gpurun_out/pmc_${W}_FETCH_SIZE gpurun_out/pmc_${W}_WRITE_SIZE $W > profiles/${R}_pmc_traffic$sfx.json
  cp gpurun_out/pmc_${W}_summary.txt profiles/${R}_pmc_summary$sfx.txt
  cp gpurun_out/prof_${R}_${W}_summary.txt profiles/${R}_kernel_stats_summary$sfx.txt
  cp "$(ls -t gpurun_out/prof_${R}_$W/*/*_kernel_stats.csv | head -1)" profiles/${R}_kernel_stats$sfx.csv
done
# (the Schur-form counter passes are a run of their own — tools/pmc_schur.sh; copied only if that run is newer than
# the kernel trace of this round, so that a round that did not repeat them does not re-label old files)
for W in kitti00 synth10k; do for V in mfma gather strip rows; do
  f=gpurun_out/pmc_schur_${W}_${V}.txt
  if [ -f $f ] && [ $f -nt gpurun_out/prof_${R}_kitti00_summary.txt ]; then cp $f profiles/${R}_pmc_schur_${W}_${V}.txt; fi
done; done
R=$R python - <<'PY'
import json, os
R = os.environ["R"]
for src, dst in (("gpurun_out/bench_kitti00.json", "profiles/%s_bench_kitti00.json" % R),
                 ("gpurun_out/bench_synth10k.json", "profiles/%s_bench_synth10k_1gpu.json" % R),
                 ("gpurun_out/bench_kitti00_float32.json", "profiles/%s_bench_kitti00_float32.json" % R)):
    line = [l for l in open(src) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(dst, "w"), indent=1)
d = json.load(open("profiles/%s_bench_kitti00.json" % R))
print({k: d[k] for k in ("value", "ms_per_step")}, d["optimize_only"]["ms_per_step"], d["reflatten"]["ms_per_step"],
      d["structure_dirty"]["ms_per_step"], d["parity"]["max_rel_chi2_diff_vs_cpu"], d["cpu_baseline"]["legs"])
for k, v in d["kernels"].items():
    if "frac" in v:
        print("%-24s %8.1f us  %9.3f %s  frac %.4f" % (k, v["avg_ms"] * 1e3, v["achieved"], v["unit"], v["frac"]))
s = json.load(open("profiles/%s_bench_synth10k_1gpu.json" % R))
print("synth10k ms_per_step", s["ms_per_step"], s["optimize_only"]["ms_per_step"], s["structure_dirty"]["ms_per_step"])
PY
