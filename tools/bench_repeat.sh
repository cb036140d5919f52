#!/bin/bash
# GPU box: ms_per_step of N bench runs (default 6), one line
N=${1:-6}; shift
for i in $(seq $N); do env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'], end=' ')"; done; echo
