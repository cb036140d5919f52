#!/bin/bash
# sweep of the ordering / supernode parameters: bash tools/sweep_ordering.sh [workload]
cd ${GRAFT_REPO_ROOT:-.}
W=${1:-kitti00}
for cfg in "24 8 0.35" "48 16 0.2" "96 16 0.2" "96 16 0.35" "192 16 0.2"; do
  set -- $cfg
  r=$(CUGO_ND_LEAF=$1 CUGO_MAX_SUPER_COLS=$2 CUGO_ZERO_FRAC=$3 timeout -k 10 200 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%.2f ms stages %d supernodes %d cold %.0f ms' % (d['ms_per_step'], d['structure']['stages'], d['structure']['supernodes'], d['cold_first_call']['optimize_ms']))")
  echo "$W leaf $1 super $2 zero $3 : $r"
done
