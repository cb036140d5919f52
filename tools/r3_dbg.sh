#!/bin/bash
mkdir -p gpurun_out
v() { echo "=== $*"; timeout -k 10 400 env $1 python tools/shard_dbg.py $2 $3 $4 $5 $6 > gpurun_out/last.log 2>&1; grep -v "^Extension\|^  File\|^$\|^Thread" gpurun_out/last.log | tail -2; if grep -q "Memory access fault\|GPU core dump" gpurun_out/last.log; then echo "FAULT: $*"; exit 1; fi; }
v CUGO_OWN_SUBTREES=0 700 12000 50000 8 2
v CUGO_OWN_SUBTREES=1 700 12000 50000 8 2
v CUGO_OWN_SUBTREES=1 3000 300000 1500000 8 2
echo "all ran"
