set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python tools/repro_stage.py 40000 6000 > gpurun_out/repro_stage.txt 2>&1 || true
cat gpurun_out/repro_stage.txt | tail -12
if grep -q "Memory access fault" gpurun_out/repro_stage.txt; then exit 1; fi
echo done
