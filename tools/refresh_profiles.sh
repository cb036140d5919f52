# GPU box: tests, the bench lines of both BASELINE shapes, rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE counter passes of the same bench command (separate --pmc passes, no
# trace domains besides --kernel-trace).  Afterwards: tools/collect_profiles.sh copies the summaries
# into profiles/.   gpurun --timeout 1100 -- bash tools/refresh_profiles.sh
set -e
R=${R:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_kitti00.json 2> gpurun_out/bench_kitti00.err
timeout -k 10 400 python bench.py --workload synth10k --steps 3 --warmup 1 > gpurun_out/bench_synth10k.json 2> gpurun_out/bench_synth10k.err
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --float32 --no-cpu-baseline > gpurun_out/bench_kitti00_float32.json 2> gpurun_out/bench_kitti00_float32.err
for W in kitti00 synth10k; do
  rm -rf gpurun_out/prof_${R}_$W
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_$W -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_${R}_$W.json 2> gpurun_out/prof_${R}_$W.err
  python tools/prof_summary.py gpurun_out/prof_${R}_$W timeline > gpurun_out/prof_${R}_${W}_summary.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${W}_$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${W}_$c -- python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_${W}_$c.json 2> gpurun_out/pmc_${W}_$c.err
  done
  python tools/pmc_summary.py gpurun_out/pmc_${W}_FETCH_SIZE gpurun_out/pmc_${W}_WRITE_SIZE > gpurun_out/pmc_${W}_summary.txt 2>&1
done
tail -2 gpurun_out/gpu_tests.log
