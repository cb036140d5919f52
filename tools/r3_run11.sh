set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/gpu_tests.log; then exit 1; fi
python tools/ab_env.py CUGO_TRIAL_EVENT 1 0 --reps 20 > gpurun_out/ab_r3_run11.txt 2>&1
cat gpurun_out/ab_r3_run11.txt
echo done
