set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CUGO_EA_PIPE=1 timeout -k 10 300 python -m pytest tests/test_gpu.py -m gpu -q -x -k "sparse_cholesky or medium_synthetic or golden" > gpurun_out/gpu_tests_eapipe.log 2>&1 || { tail -20 gpurun_out/gpu_tests_eapipe.log; exit 1; }
tail -1 gpurun_out/gpu_tests_eapipe.log
{
python tools/ab_env.py CUGO_EA_PIPE 1 0 --reps 30
python tools/ab_env.py CUGO_EA_PIPE 1 0 --workload synth10k --reps 8
} > gpurun_out/ab_r3_run24.txt 2>&1
cat gpurun_out/ab_r3_run24.txt
echo done
