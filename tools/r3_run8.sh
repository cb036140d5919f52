# GPU box: run-to-run reproducibility probe over the switches of this round, then (if clean) the GPU suite
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/repro_check.py synth10k "" CUGO_EA_UNITS=0 CUGO_UPLOAD_THREAD=0 CUGO_TRIAL_EVENT=0 CUGO_SPECULATE=0 CUGO_HSC_MFMA=0 CUGO_ASM_FRONTS=0 CUGO_HSC_XCD=0 > gpurun_out/repro_synth10k.txt 2>&1 || true
cat gpurun_out/repro_synth10k.txt
CUGO_POISON_ALLOC=1 timeout -k 10 300 python tools/repro_check.py synth10k "" > gpurun_out/repro_synth10k_poison.txt 2>&1 || true
cat gpurun_out/repro_synth10k_poison.txt
CUGO_POISON_ALLOC=1 timeout -k 10 300 python tools/repro_check.py kitti00 "" > gpurun_out/repro_kitti00_poison.txt 2>&1 || true
cat gpurun_out/repro_kitti00_poison.txt
if grep -q "Memory access fault" gpurun_out/repro_*.txt; then exit 1; fi
echo done
