#!/bin/bash
# GPU box: the bench line's ms_per_step for several values of a PROCESS-wide environment variable (one process per value,
# alternating, three rounds):   bash tools/env_procs.sh VAR "v1 v2" [workload]
VAR=$1; VALS=$2; W=${3:-kitti00}
for R in 1 2 3; do
 for V in $VALS; do
  env $VAR=$V python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/env_procs.json 2> gpurun_out/env_procs.err
  echo "$W $VAR=$V  $(python -c "import json;d=json.loads([l for l in open('gpurun_out/env_procs.json') if l.startswith('{')][0]);print('ms_per_step %.3f' % d['ms_per_step'])")"
 done
done
