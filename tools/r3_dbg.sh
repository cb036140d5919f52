#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do
  echo "=== CUGO_HSC_ROWS=$v" 
  CUGO_HSC_ROWS=$v timeout -k 10 200 python -m pytest tests/test_gpu.py -x -q -k "degenerate_fixed_sets" 2>&1 | grep -v "^Extension\|^  File\|^$" | tail -15
done
