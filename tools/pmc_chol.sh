# GPU box: instruction-mix / LDS / wait counters of the sparse-Cholesky kernels (separate --pmc passes,
# kernel trace only).   gpurun --timeout 900 -- bash tools/pmc_chol.sh ; summary in gpurun_out/pmc_chol_summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/pmc_chol_summary.txt
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmcc_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcc_$tag -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmcc_$tag.json 2> gpurun_out/pmcc_$tag.err || exit 1
  python tools/pmc_summary.py gpurun_out/pmcc_$tag | grep "k_up_\|k_backward\|==" >> gpurun_out/pmc_chol_summary.txt 2>&1
  rm -rf gpurun_out/pmcc_$tag
done
cat gpurun_out/pmc_chol_summary.txt
