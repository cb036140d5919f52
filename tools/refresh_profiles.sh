set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_kitti00.json 2> gpurun_out/bench_kitti00.err
timeout -k 10 300 python bench.py --workload synth10k --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_synth10k.json 2> gpurun_out/bench_synth10k.err
rm -rf gpurun_out/prof_r01 && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r01_bench.json 2> gpurun_out/prof_r01_err.log
python tests/prof_summary.py gpurun_out/prof_r01 timeline > gpurun_out/prof_r01_summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
done
python tests/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.txt 2>&1
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --float32 > gpurun_out/bench_kitti00_float32.json 2> gpurun_out/bench_kitti00_float32.err
