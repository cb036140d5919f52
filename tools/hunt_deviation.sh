# GPU box: is this a box (and a time) at which the rare run-to-run deviation shows?  If so:
# which kind of separation between the kernels of the factorisation makes it go away?
#   gpurun --timeout 1150 -- bash tools/hunt_deviation.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # name, runs, env...
  name=$1; n=$2; shift 2
  env "$@" timeout -k 10 400 python tools/repro_medium.py $n "" > gpurun_out/hunt_$name.txt 2>&1 || true
  echo "$name: $(grep -v '^    it \|^  run\|^    first' gpurun_out/hunt_$name.txt | cut -c1-160)"
}
run control1 800 CUGO_X=0
if grep -q "deviating 0 " gpurun_out/hunt_control1.txt; then echo "clean box"; exit 0; fi
run gap 2500 CUGO_DEBUG_GAP=1
run control2 800 CUGO_X=0
run serialize3 2500 AMD_SERIALIZE_KERNEL=3
run control3 800 CUGO_X=0
run serialize1 2500 AMD_SERIALIZE_KERNEL=1
run serialize2 2500 AMD_SERIALIZE_KERNEL=2
run control4 800 CUGO_X=0
run panel16_0 2500 CUGO_PANEL16=0
run hsc_mfma_0 2500 CUGO_HSC_MFMA=0
run control5 800 CUGO_X=0
if grep -q "Memory access fault" gpurun_out/hunt_*.txt; then exit 1; fi
echo done
