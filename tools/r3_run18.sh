set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python tools/repro_small.py 192 "" CUGO_HSC_MFMA=0 CUGO_ASM_FRONTS=0 CUGO_PANEL16=0 CUGO_SPECULATE=0 CUGO_TRIAL_POLL=0 CUGO_ASYNC_STRUCTURE=0,CUGO_UPLOAD_THREAD=0 > gpurun_out/repro_small.txt 2>&1 || true
cat gpurun_out/repro_small.txt
CUGO_POISON_ALLOC=1 timeout -k 10 300 python tools/repro_small.py 96 "" > gpurun_out/repro_small_poison.txt 2>&1 || true
cat gpurun_out/repro_small_poison.txt
if grep -q "Memory access fault" gpurun_out/repro_small*.txt; then exit 1; fi
echo done
