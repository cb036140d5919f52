set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python tools/ab_env.py CUGO_TWO_PHASE_MIN_TILES 128 170 200 260 --reps 20
python tools/ab_env.py CUGO_TILE32_MAX_TILES 64 100 40 --reps 20
python tools/ab_env.py CUGO_TWO_PHASE_MIN_TILES 128 200 400 --reps 6 --workload synth10k
python tools/ab_env.py CUGO_TILE32_MAX_TILES 64 100 40 --reps 6 --workload synth10k
} > gpurun_out/ab_r3_run14.txt 2>&1
cat gpurun_out/ab_r3_run14.txt
