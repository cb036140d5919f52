set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python tools/repro_medium.py 300 "" CUGO_TWO_PHASE_MIN_TILES=1,CUGO_TILE32_MAX_TILES=0 CUGO_EA_PIPE=1 CUGO_EA_PIPE=1,CUGO_TWO_PHASE_MIN_TILES=1,CUGO_TILE32_MAX_TILES=0 F32=1 CUGO_PANEL16=0 CUGO_HSC_MFMA=0 CUGO_SPECULATE=0 CUGO_TRIAL_POLL=0 > gpurun_out/repro_medium.txt 2>&1 || true
cat gpurun_out/repro_medium.txt
if grep -q "Memory access fault" gpurun_out/repro_medium.txt; then exit 1; fi
echo done
