# GPU box: the harness of the round-3 hunt for a rare run-to-run deviation (cause found: a race
# between the two waves of a wide panel in k_up_potrf, fixed).  Kept as a regression harness: self-tests of the card
# (per-CU bit equality, cross-XCD visibility), a control (800 fresh optimisers on the 400-pose graph, bitwise), and — if
# the control deviates — the autopsy of a deviating run, interleaved blocks of switches, fault injection.
# The sharper detector is the 10k-pose graph:  python tools/repro_medium.py 300 --10k ""  /  python tools/autopsy.py 600 --10k
#   gpurun --timeout 1150 -- bash tools/hunt_deviation.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/hunt_*.txt
block() { # name, runs, env...
  name=$1; n=$2; shift 2
  env "$@" timeout -k 10 300 python tools/repro_medium.py $n "" >> gpurun_out/hunt_$name.txt 2>&1 || true
}
HOOKS=$GRAFT_REPO_ROOT/cuda-bundle-adjustment_amd/libcugo_hip_hooks.so  # the build that carries the kernels' diagnosis hooks
[ -f $HOOKS ] || make -C cuda-bundle-adjustment_amd HOOKS=1 -j16 -s
hipcc --offload-arch=gfx950 -O2 tools/mfma_selftest.hip -o /tmp/mfma_selftest 2> /dev/null
hipcc --offload-arch=gfx950 -O2 tools/coherence_selftest.hip -o /tmp/coherence_selftest 2> /dev/null
selftest() {
  timeout -k 10 200 /tmp/mfma_selftest 240 20000 > gpurun_out/hunt_selftest_$1.txt 2>&1 || true; tail -12 gpurun_out/hunt_selftest_$1.txt | cut -c1-200
  timeout -k 10 200 /tmp/coherence_selftest $2 > gpurun_out/hunt_coherence_$1.txt 2>&1 || true; tail -30 gpurun_out/hunt_coherence_$1.txt | cut -c1-200
}
selftest before 10000
block control 800 CUGO_X=0
grep "deviating" gpurun_out/hunt_control.txt | cut -c1-160
if grep -q "deviating 0 " gpurun_out/hunt_control.txt; then echo "clean box"; exit 0; fi
# where a deviating run differs from its twin (fronts, W, L21, x after every factorisation)
timeout -k 10 400 python tools/autopsy.py 6000 > gpurun_out/hunt_autopsy.txt 2>&1 || true
cut -c1-220 gpurun_out/hunt_autopsy.txt | head -150
for round in 1 2 3 4 5 6; do
  block default 400 CUGO_X=0
  block panel16_0 400 CUGO_PANEL16=0
  block hsc_mfma_0 400 CUGO_HSC_MFMA=0
  block hooks_build 400 CUGO_LIB=$HOOKS
  block kernel_acquire 400 CUGO_KERNEL_ACQUIRE=1 CUGO_LIB=$HOOKS
  block kernel_release 400 CUGO_KERNEL_ACQUIRE=2 CUGO_LIB=$HOOKS
  block wave_waits 400 CUGO_KERNEL_ACQUIRE=4 CUGO_LIB=$HOOKS
  block serialize3 400 AMD_SERIALIZE_KERNEL=3
  block hash 400 CUGO_DEBUG_HASH=/tmp/hunt_hash.txt
  block hash_ends 400 CUGO_DEBUG_HASH=/tmp/hunt_hash2.txt CUGO_DEBUG_HASH_LEVELS=0
  block xcd_affinity_0 400 CUGO_XCD_AFFINITY=0
  block round2_paths 400 CUGO_PANEL16=0 CUGO_HSC_MFMA=0 CUGO_ASM_FRONTS=0 CUGO_TRIAL_POLL=0 CUGO_SPECULATE=0 CUGO_EA_LDS=0
  echo "round $round done"
done
for n in default panel16_0 hsc_mfma_0 hooks_build kernel_acquire kernel_release wave_waits serialize3 hash hash_ends xcd_affinity_0 round2_paths; do
  echo "$n: $(grep -c 'first chi2 difference' gpurun_out/hunt_$n.txt || true) deviating of $(grep -c ' runs ' gpurun_out/hunt_$n.txt)x400"
done
grep -h "first differing" gpurun_out/hunt_hash.txt gpurun_out/hunt_hash_ends.txt | cut -c1-300 || true
selftest after 60000
# the factorisation alone, back to back with changing inputs (a value left over from the previous call shows)
timeout -k 10 300 python tools/repro_chain.py 1500 "" CUGO_PANEL16=0 > gpurun_out/hunt_chain.txt 2>&1 || true
cut -c1-200 gpurun_out/hunt_chain.txt
timeout -k 10 300 python tools/repro_chain.py 1500 --schur "" CUGO_PANEL16=0 > gpurun_out/hunt_chain_schur.txt 2>&1 || true
cut -c1-200 gpurun_out/hunt_chain_schur.txt
if grep -q "Memory access fault" gpurun_out/hunt_*.txt; then exit 1; fi
echo done
