# GPU box: is this a box on which the rare deviation shows?  If so: localise it with the in-stream checksums and
# with every stage alone.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/repro_medium.py 1000 "" > gpurun_out/repro_medium6a.txt 2>&1 || true
grep -v "^    it \|^  run" gpurun_out/repro_medium6a.txt | cut -c1-200
if grep -q "deviating 0 " gpurun_out/repro_medium6a.txt; then echo "clean box"; exit 0; fi
export CUGO_DEBUG_HASH=/tmp/cugo_hashes.txt
timeout -k 10 500 python tools/repro_medium.py 5000 "" > gpurun_out/repro_medium6.txt 2>&1 || true
grep -v "^    it " gpurun_out/repro_medium6.txt | cut -c1-300 | head -70
unset CUGO_DEBUG_HASH
timeout -k 10 300 python tools/repro_stage.py 40000 5000 > gpurun_out/repro_stage6.txt 2>&1 || true
tail -6 gpurun_out/repro_stage6.txt
if grep -q "Memory access fault" gpurun_out/repro_medium6*.txt gpurun_out/repro_stage6.txt; then exit 1; fi
echo done
