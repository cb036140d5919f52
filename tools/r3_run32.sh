set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/delay_check.py > gpurun_out/delay_check.txt 2>&1 || true
cat gpurun_out/delay_check.txt
if grep -q "Memory access fault" gpurun_out/delay_check.txt; then exit 1; fi
