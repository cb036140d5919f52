set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/gpu_tests.log; then exit 1; fi
{
python tools/ab_env.py CUGO_TRIAL_POLL 1 0 --reps 30
python tools/ab_env.py CUGO_TRIAL_POLL 1 0 --reps 8 --workload synth10k
} > gpurun_out/ab_r3_run12.txt 2>&1
cat gpurun_out/ab_r3_run12.txt
timeout -k 10 300 python bench.py --workload localba --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_localba.json 2> gpurun_out/bench_localba.err
CUGO_TRIAL_POLL=0 timeout -k 10 300 python bench.py --workload localba --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_localba_nopoll.json 2> gpurun_out/bench_localba_nopoll.err
python -c "
import json
for f in ['bench_localba','bench_localba_nopoll']:
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'])"
echo done
