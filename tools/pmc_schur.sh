# GPU box: FETCH_SIZE / WRITE_SIZE / TCC hit-miss counter passes (separate --pmc runs, --kernel-trace only) of the
# Schur-complement kernels in their four forms, both shapes -> gpurun_out/pmc_schur_<shape>_<form>.txt
#   gpurun --timeout 1100 -- bash tools/pmc_schur.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# the H-side of the Schur complement in its four forms (default: gather + matrix cores; gather on the vector
# lanes; row strips; whole block rows): traffic and L2 behaviour per launch, both shapes
for W in kitti00 synth10k; do
  for V in mfma gather strip rows; do
    case $V in mfma) ENVV="CUGO_X=0";; gather) ENVV="CUGO_HSC_MFMA=0";; strip) ENVV="CUGO_HSC_STRIP=1";; rows) ENVV="CUGO_HSC_ROWS=1";; esac
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      tag=${W}_${V}_$(echo $c | tr ' ' '_')
      rm -rf gpurun_out/pmc_schur_$tag
      export $ENVV
      timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_schur_$tag -- python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_schur_$tag.json 2> gpurun_out/pmc_schur_$tag.err
      unset CUGO_HSC_STRIP CUGO_HSC_ROWS CUGO_HSC_MFMA CUGO_X
    done
    python tools/pmc_summary.py gpurun_out/pmc_schur_${W}_${V}_FETCH_SIZE gpurun_out/pmc_schur_${W}_${V}_WRITE_SIZE gpurun_out/pmc_schur_${W}_${V}_TCC_HIT_sum_TCC_MISS_sum 2>&1 | grep "k_hsc\|k_schur\|k_inv_hll\|==" > gpurun_out/pmc_schur_${W}_${V}.txt
  done
done
