set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1050 python tools/repro_medium.py 1200 "" CUGO_DEBUG_ZERO_LDS=1 "" CUGO_DEBUG_ZERO_LDS=1 > gpurun_out/repro_medium4.txt 2>&1 || true
grep -v "^    it\|^  run" gpurun_out/repro_medium4.txt | cut -c1-200
if grep -q "Memory access fault" gpurun_out/repro_medium4.txt; then exit 1; fi
echo done
