# GPU box: the GPU suite with guard zones + NaN poisoning of every device buffer (CUGO_POISON_ALLOC=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CUGO_POISON_ALLOC=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests_poison.log 2>&1
echo "rc $?"
tail -3 gpurun_out/gpu_tests_poison.log
grep -n "guard zone\|FAILED\|Aborted\|Memory access fault" gpurun_out/gpu_tests_poison.log | head -20
if grep -q "Memory access fault" gpurun_out/gpu_tests_poison.log; then exit 1; fi
echo done
