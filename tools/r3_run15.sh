set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/gpu_tests.log; then exit 1; fi
{
python tools/ab_env.py CUGO_TWO_PHASE_MIN_TILES 128 260 400 600 100000 --reps 20
python tools/ab_env.py CUGO_TWO_PHASE_MIN_TILES 128 400 800 1600 100000 --reps 6 --workload synth10k
} > gpurun_out/ab_r3_run15.txt 2>&1
cat gpurun_out/ab_r3_run15.txt
for W in kitti00 synth10k; do
  rm -rf gpurun_out/prof_new_$W
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_new_$W -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_new_$W.json 2> gpurun_out/prof_new_$W.err
  python tools/prof_summary.py gpurun_out/prof_new_$W > gpurun_out/prof_new_${W}_summary.txt 2>&1 || true
  grep "k_backward\|k_up_" gpurun_out/prof_new_${W}_summary.txt
done
CUGO_LIB=cuda-bundle-adjustment_amd/libcugo_hip_stamps.so CUGO_DEBUG_STAMPS=1 timeout -k 10 300 python tools/stamps_run.py > gpurun_out/stamps_now.txt 2>&1 || true
grep "kernel 3\|kernel 0" gpurun_out/stamps_now.txt
echo done
