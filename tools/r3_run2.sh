#!/bin/bash
# round 3: whole GPU suite, A/B of a switch (args: VAR a b), kernel-trace timeline of the bench command
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
if [ -n "$1" ]; then timeout -k 10 200 python tools/ab_env.py $1 $2 $3 --reps 20 > gpurun_out/r3_ab2.log 2>&1; cat gpurun_out/r3_ab2.log; fi
if [ "$4" = "dirty" ]; then timeout -k 10 300 python tools/ab_env.py $1 $2 $3 --reps 15 --dirty > gpurun_out/r3_ab2_dirty.log 2>&1; cat gpurun_out/r3_ab2_dirty.log; timeout -k 10 400 python tools/ab_env.py $1 $2 $3 --reps 5 --dirty --workload synth10k > gpurun_out/r3_ab2_dirty10k.log 2>&1; cat gpurun_out/r3_ab2_dirty10k.log
elif [ -n "$4" ]; then timeout -k 10 400 python tools/ab_env.py $1 $2 $3 --reps 6 --workload $4 > gpurun_out/r3_ab2_$4.log 2>&1; cat gpurun_out/r3_ab2_$4.log; fi
rm -rf gpurun_out/prof_r03_kitti00
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_kitti00 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_r03_kitti00.json 2> gpurun_out/prof_r03_kitti00.err
python tools/prof_summary.py gpurun_out/prof_r03_kitti00 timeline > gpurun_out/prof_r03_kitti00_summary.txt 2>&1
head -40 gpurun_out/prof_r03_kitti00_summary.txt
