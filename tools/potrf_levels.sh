#!/bin/bash
# GPU box: duration of every k_up_potrf launch of one factorisation (rocprofv3 kernel trace of the bench), per value
# of an environment switch.   bash tools/potrf_levels.sh VAR "v1 v2" [workload]
VAR=${1:-CUGO_X}; VALS=${2:-0}; W=${3:-kitti00}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in $VALS; do
  rm -rf gpurun_out/prof_x
  env $VAR=$V rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_x.json 2> gpurun_out/prof_x.err
  echo "== $W $VAR=$V $(python -c "import json;d=json.loads([l for l in open('gpurun_out/prof_x.json') if l.startswith('{')][0]);print('ms_per_step %.3f chi2_last %.6f' % (d['ms_per_step'], d['chi2'][-1]))")"
  python tools/prof_summary.py gpurun_out/prof_x timeline | grep -E "k_up_potrf " | head -19 | awk '{printf "%s ", $6} END {print ""}'
  python tools/prof_summary.py gpurun_out/prof_x | grep -E "k_up_potrf|k_up_trsyrk|k_backward"
done
