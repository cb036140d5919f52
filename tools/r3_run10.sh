# GPU box: reproducibility probe with the device idling between the runs (what the suite's failing case did)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export REPRO_SLEEP=4
timeout -k 10 900 python tools/repro_check.py synth10k "" "" CUGO_TRIAL_EVENT=0 CUGO_SPECULATE=0 CUGO_HSC_MFMA=0 CUGO_ASM_FRONTS=0 CUGO_PANEL16=0 > gpurun_out/repro_sleep_synth10k.txt 2>&1 || true
cut -c1-150 gpurun_out/repro_sleep_synth10k.txt | grep -v "chi trace"
if grep -q "Memory access fault" gpurun_out/repro_*.txt; then exit 1; fi
echo done
