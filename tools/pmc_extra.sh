# one-off: L2 hit / miss and request counters of the bench kernels (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmcx_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcx_$tag -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcx_$tag.json 2> gpurun_out/pmcx_$tag.err
  python tools/pmc_summary.py gpurun_out/pmcx_$tag > gpurun_out/pmcx_$tag.txt 2>&1
done
cat gpurun_out/pmcx_*.txt | grep "k_hsc\|k_build_edges\|k_schur\|k_backsubst\|k_errors\|==" 
