#!/bin/bash
# GPU box: timeline of one iteration with / without the side stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in 0 1; do
  rm -rf gpurun_out/prof_side
  env CUGO_SIDE_STREAM=$V rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_side -- python bench.py --workload ${1:-kitti00} --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_side.json 2> gpurun_out/prof_side.err
  echo "== CUGO_SIDE_STREAM=$V"
  python tools/iter_timeline.py gpurun_out/prof_side 'k_errors|k_sum|k_build|k_hsc|k_assemble|k_backsubst'
done
