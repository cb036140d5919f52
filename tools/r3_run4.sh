# GPU box: A/B of the trial-event wait, CUGO_INIT_TIMING laps of the new-graph regime, then the Schur counter passes
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/ab_env.py CUGO_TRIAL_EVENT 1 0 --reps 30 > gpurun_out/ab_trial_event.txt 2>&1
python tools/ab_env.py CUGO_TRIAL_EVENT 1 0 --workload synth10k --reps 8 >> gpurun_out/ab_trial_event.txt 2>&1
cat gpurun_out/ab_trial_event.txt
CUGO_INIT_TIMING=1 python tools/ab_env.py CUGO_TRIAL_EVENT 1 --dirty --reps 3 > gpurun_out/init_timing_kitti00.txt 2>&1
CUGO_INIT_TIMING=1 python tools/ab_env.py CUGO_TRIAL_EVENT 1 --dirty --reps 3 --workload synth10k > gpurun_out/init_timing_synth10k.txt 2>&1
tail -40 gpurun_out/init_timing_kitti00.txt
bash tools/pmc_schur.sh
