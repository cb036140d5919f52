#!/bin/bash
# round 3: GPU suite + A/Bs given as triples "VAR a b" (kitti00, and synth10k for the first)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
grep -q "Memory access fault\|GPU core dump" gpurun_out/gpu_tests.log && { echo "GPU FAULT in the test suite"; exit 1; }
first=1
while [ $# -ge 3 ]; do
  timeout -k 10 200 python tools/ab_env.py $1 $2 $3 --reps 20 2>&1 | tail -2
  if [ $first = 1 ]; then timeout -k 10 400 python tools/ab_env.py $1 $2 $3 --reps 6 --workload synth10k 2>&1 | tail -2; first=0; fi
  shift 3
done
