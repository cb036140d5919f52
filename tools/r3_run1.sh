#!/bin/bash
# round 3: tests of the Cholesky paths, A/B of the 16-column potrf, cycle stamps of both forms
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu.py tests/test_boundary.py -x -q -m gpu -k "sparse_cholesky or lm_trajectory or medium_synthetic or kitti07 or seam or schur_complement" > gpurun_out/r3_t1.log 2>&1 || tail -40 gpurun_out/r3_t1.log
tail -3 gpurun_out/r3_t1.log
python tools/ab_env.py CUGO_PANEL16 1 0 --reps 20 > gpurun_out/r3_ab1.log 2>&1 && cat gpurun_out/r3_ab1.log
rm -f gpurun_out/r3_stamps1.log
for v in 1 0; do
  echo "== stamps CUGO_PANEL16=$v" >> gpurun_out/r3_stamps1.log
  CUGO_PANEL16=$v CUGO_LIB=$PWD/cuda-bundle-adjustment_amd/libcugo_hip_stamps.so CUGO_DEBUG_STAMPS=1 python tools/stamps_run.py >> gpurun_out/r3_stamps1.log 2>&1
done
cat gpurun_out/r3_stamps1.log
