# GPU box: the GPU suite with guard zones and every fresh floating-point buffer filled with NaNs (=1), then with
# finite garbage (=2: 32.5 — what max / min / comparisons do not swallow)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 1 2; do
  CUGO_POISON_ALLOC=$m timeout -k 10 500 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests_poison$m.log 2>&1
  echo "poison $m rc $?: $(tail -1 gpurun_out/gpu_tests_poison$m.log)"
  grep -n "guard zone\|^FAILED\|Aborted\|Memory access fault" gpurun_out/gpu_tests_poison$m.log | head -20
  if grep -q "Memory access fault" gpurun_out/gpu_tests_poison$m.log; then exit 1; fi
done
echo done
