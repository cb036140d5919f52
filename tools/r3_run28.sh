set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1050 python tools/repro_medium.py 1200 "" CUGO_FUSE_T=0 CUGO_HSC_MFMA=0 CUGO_PANEL16=0 CUGO_ASM_FRONTS=0 CUGO_ASYNC_STRUCTURE=0,CUGO_UPLOAD_THREAD=0 CUGO_EA_LDS=0 CUGO_TILE32_MAX_TILES=0 CUGO_SPECULATE=0,CUGO_FUSE_T=0 "" > gpurun_out/repro_medium3.txt 2>&1 || true
grep -v "^    it\|^  run" gpurun_out/repro_medium3.txt | cut -c1-260
if grep -q "Memory access fault" gpurun_out/repro_medium3.txt; then exit 1; fi
echo done
