# GPU box: the tail of the GPU suite (threaded shard tests -> full-size 10k -> soak) four times, to see whether the
# run-to-run differences seen twice in full suite runs recur
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout -k 10 400 python -m pytest tests/test_gpu.py -m gpu -q -k "rank_owned or eight_shards or two_shards or synth10k_full or native_comm or two_ranks or repeated_new" > gpurun_out/suite_tail_$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/suite_tail_$i.log)"
  if grep -q "Memory access fault" gpurun_out/suite_tail_$i.log; then exit 1; fi
  grep -n "AssertionError\|^E  " gpurun_out/suite_tail_$i.log | head -8
done
echo done
