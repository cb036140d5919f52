# GPU box: the full GPU suite with the new defaults, then A/Bs of the matrix-core Schur kernels and the one-pass assembly
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/gpu_tests.log; then exit 1; fi
{
python tools/ab_env.py CUGO_HSC_MFMA 1 2 0 --reps 30
python tools/ab_env.py CUGO_HSC_XCD 1 0 --reps 30
python tools/ab_env.py CUGO_ASM_FRONTS 1 0 --reps 30
python tools/ab_env.py CUGO_HSC_MFMA 1 2 0 --workload synth10k --reps 8
python tools/ab_env.py CUGO_HSC_XCD 1 0 --workload synth10k --reps 8
python tools/ab_env.py CUGO_ASM_FRONTS 1 0 --workload synth10k --reps 8
} > gpurun_out/ab_r3_run5.txt 2>&1
cat gpurun_out/ab_r3_run5.txt
for W in kitti00 synth10k; do
  rm -rf gpurun_out/prof_new_$W
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_new_$W -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_new_$W.json 2> gpurun_out/prof_new_$W.err
  python tools/prof_summary.py gpurun_out/prof_new_$W timeline > gpurun_out/prof_new_${W}_summary.txt 2>&1 || true
done
rm -rf gpurun_out/prof_dirty
export CUGO_NO_STRUCTURE_REUSE=1 CUGO_NO_FLATTEN_REUSE=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dirty -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_dirty.json 2> gpurun_out/prof_dirty.err
python tools/prof_summary.py gpurun_out/prof_dirty timeline > gpurun_out/prof_dirty_summary.txt 2>&1 || true
echo done
