set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
for W in kitti00; do
  rm -rf gpurun_out/prof_new_$W
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_new_$W -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_new_$W.json 2> gpurun_out/prof_new_$W.err
  python tools/prof_summary.py gpurun_out/prof_new_$W > gpurun_out/prof_new_${W}_summary.txt 2>&1 || true
  grep "k_up_potrf\|k_backward" gpurun_out/prof_new_${W}_summary.txt
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti00 ms_per_step', d['ms_per_step'])"
echo done
