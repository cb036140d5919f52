# GPU box: the medium graph optimised N times while ANOTHER process keeps the card busy (tools/mfma_selftest: every
# CU full of fp64 matrix-core waves) — workgroup placement, timing and queue scheduling all differ from the idle
# card's.  Run-to-run deviations are counted as in tools/repro_medium.py.
#   gpurun --timeout 900 -- bash tools/contention_check.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/mfma_selftest.hip -o /tmp/mfma_selftest 2> /dev/null
timeout -k 10 200 python tools/repro_medium.py 300 "" > gpurun_out/contention_idle.txt 2>&1 || true
grep "deviating" gpurun_out/contention_idle.txt | cut -c1-200
timeout -k 10 400 /tmp/mfma_selftest 100000 20000 > gpurun_out/contention_load.txt 2>&1 &
LOAD=$!
sleep 3
timeout -k 10 300 python tools/repro_medium.py 300 "" > gpurun_out/contention_busy.txt 2>&1 || true
grep "deviating" gpurun_out/contention_busy.txt | cut -c1-300
timeout -k 10 300 python tools/repro_medium.py 300 CUGO_PANEL16=0 CUGO_HSC_MFMA=0 > gpurun_out/contention_busy2.txt 2>&1 || true
grep "deviating" gpurun_out/contention_busy2.txt | cut -c1-300
kill $LOAD 2> /dev/null || true
wait $LOAD 2> /dev/null || true
tail -3 gpurun_out/contention_load.txt | cut -c1-200
if grep -q "Memory access fault" gpurun_out/contention_*.txt; then exit 1; fi
echo done
