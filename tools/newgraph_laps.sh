#!/bin/bash
# GPU box: the new-graph regime (flatten + structure rebuilt in every call) at several widths of the host pool
# (CUGO_HOST_THREADS; the pool is created once per process, so one process per width), then the per-section laps.
#   bash tools/newgraph_laps.sh [kitti00|synth10k]
WL=${1:-kitti00}
echo "nproc $(nproc)  affinity $(python -c 'import os;print(len(os.sched_getaffinity(0)))')  cpu $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
for T in 8 16 24 32 48 64; do
  echo "== CUGO_HOST_THREADS=$T"
  CUGO_HOST_THREADS=$T python tools/ab_env.py CUGO_UNUSED_SWITCH 0 --dirty --workload $WL --reps 12 2>&1 | tail -1
done
echo "== default width, laps of one new-graph call"
if [ "$WL" = "kitti00" ]; then python tools/cold_small.py 2>&1 | tail -45; else python tools/cold_10k.py 2>&1 | tail -45; fi
