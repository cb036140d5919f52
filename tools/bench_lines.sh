# GPU box: the three bench lines only (the committed profiles/ supply the rocprof / PMC fields)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_kitti00.json 2> gpurun_out/bench_kitti00.err
timeout -k 10 400 python bench.py --workload synth10k --steps 3 --warmup 1 > gpurun_out/bench_synth10k.json 2> gpurun_out/bench_synth10k.err
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --float32 --no-cpu-baseline > gpurun_out/bench_kitti00_float32.json 2> gpurun_out/bench_kitti00_float32.err
python -c "
import json
for f in ['kitti00','synth10k','kitti00_float32']:
    d=json.loads(open('gpurun_out/bench_%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], {k: v for k, v in d['config'].items() if k.startswith('regime_')})"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
