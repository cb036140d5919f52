# GPU box: the reference chi2 trace of the 10k-pose graph (for the run-to-run difference seen once in the suite),
# the GPU suite, A/B of the combined F11 / child loads
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/repro_check.py synth10k "" > gpurun_out/repro_synth10k.txt 2>&1 || true
cat gpurun_out/repro_synth10k.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/gpu_tests.log; then exit 1; fi
{
python tools/ab_env.py CUGO_EA_UNITS 1 0 --reps 30
python tools/ab_env.py CUGO_EA_UNITS 1 0 --workload synth10k --reps 8
python tools/ab_env.py CUGO_UPLOAD_THREAD 1 0 --dirty --reps 15
python tools/ab_env.py CUGO_UPLOAD_THREAD 1 0 --dirty --reps 5 --workload synth10k
} > gpurun_out/ab_r3_run9.txt 2>&1
cat gpurun_out/ab_r3_run9.txt
CUGO_INIT_TIMING=1 python tools/ab_env.py CUGO_UPLOAD_THREAD 1 --dirty --reps 3 > gpurun_out/init_timing_kitti00.txt 2>&1
CUGO_INIT_TIMING=1 python tools/ab_env.py CUGO_UPLOAD_THREAD 1 --dirty --reps 3 --workload synth10k > gpurun_out/init_timing_synth10k.txt 2>&1
tail -46 gpurun_out/init_timing_kitti00.txt
echo done
