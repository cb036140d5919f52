set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -1
python tools/ab_env.py CUGO_EA_PIPE 1 0 --reps 30
python tools/ab_env.py CUGO_EA_PIPE 1 0 --workload synth10k --reps 8
echo done
