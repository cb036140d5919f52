#!/bin/bash
# GPU box: rocprofv3 kernel averages of the bench for several values of ONE environment switch (one process each).
#   bash tools/ab_rocprof.sh VAR "v1 v2 ..." "kernel-regex" [workloads]
VAR=$1; VALS=$2; PAT=$3; WLS=${4:-"kitti00 synth10k"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in $WLS; do
 for V in $VALS; do
  rm -rf gpurun_out/prof_ab
  env $VAR=$V rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_ab.json 2> gpurun_out/prof_ab.err
  echo "== $W $VAR=$V  $(python -c "import json;d=json.loads([l for l in open('gpurun_out/prof_ab.json') if l.startswith('{')][0]);print('ms_per_step %.3f chi2_last %.6f' % (d['ms_per_step'], d['chi2'][-1]))")"
  python tools/prof_summary.py gpurun_out/prof_ab | grep -E "$PAT"
 done
done
