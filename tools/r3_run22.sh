set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CUGO_POISON_ALLOC=1 timeout -k 10 300 python -m pytest tests/test_gpu.py -m gpu -q -k "degenerate or schur or float32" > gpurun_out/gpu_tests_poison_sub.log 2>&1 || { tail -20 gpurun_out/gpu_tests_poison_sub.log; exit 1; }
tail -1 gpurun_out/gpu_tests_poison_sub.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
for W in kitti00; do
  rm -rf gpurun_out/prof_new_$W
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_new_$W -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_new_$W.json 2> gpurun_out/prof_new_$W.err
  python tools/prof_summary.py gpurun_out/prof_new_$W > gpurun_out/prof_new_${W}_summary.txt 2>&1 || true
  grep "k_hsc" gpurun_out/prof_new_${W}_summary.txt
done
python tools/ab_env.py CUGO_HSC_MFMA 1 2 --reps 20
echo done
