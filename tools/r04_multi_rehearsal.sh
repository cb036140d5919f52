set -x
python -m pytest tests/test_gpu.py -m gpu -x -q -k "native_comm or eight_shards or rank_owned or two_shards or sharded" > gpurun_out/r04_multi_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_multi_tests.log
tail -5 gpurun_out/r04_multi_tests.log
# (ii) two real processes on ONE GPU through the gloo fallback, rank-owned subtrees forced
CUGO_OWN_SUBTREES=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload synth10k --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r04_2proc_gloo_owned_synth10k.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_2proc_gloo_owned_synth10k.txt
CUGO_OWN_SUBTREES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --workload kitti00 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r04_2proc_gloo_owned_kitti00.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_2proc_gloo_owned_kitti00.txt
# (iii) refused init: both ranks on device 0 -> RCCL refuses -> every rank must exit non-zero, nobody hangs
CUGO_BENCH_ASSUME_DEVICES=1 CUGO_BENCH_COMM_TIMEOUT=40 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --workload localba --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r04_2proc_refused_init.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_2proc_refused_init.txt
# a rank that never enters the init: the other one must not wait forever
CUGO_BENCH_ASSUME_DEVICES=1 CUGO_BENCH_HANG_RANK=1 CUGO_BENCH_COMM_TIMEOUT=30 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 2 --workload localba --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r04_2proc_hung_rank.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_2proc_hung_rank.txt
tail -3 gpurun_out/r04_2proc_*.txt
