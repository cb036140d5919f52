set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python tools/repro_medium.py 2500 "" CUGO_TRIAL_POLL=0 CUGO_TRIAL_POLL=0,CUGO_TRIAL_EVENT=0 > gpurun_out/repro_medium2.txt 2>&1 || true
cat gpurun_out/repro_medium2.txt
if grep -q "Memory access fault" gpurun_out/repro_medium2.txt; then exit 1; fi
echo done
