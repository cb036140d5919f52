set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/repro_medium.py 1500 "" > gpurun_out/repro_medium5a.txt 2>&1 || true
grep -v "^    it \|^  run" gpurun_out/repro_medium5a.txt | cut -c1-200
export CUGO_DEBUG_HASH=/tmp/cugo_hashes.txt
timeout -k 10 600 python tools/repro_medium.py 4000 "" > gpurun_out/repro_medium5.txt 2>&1 || true
grep -v "^    it " gpurun_out/repro_medium5.txt | cut -c1-260 | head -60
if grep -q "Memory access fault" gpurun_out/repro_medium5*.txt; then exit 1; fi
echo done
