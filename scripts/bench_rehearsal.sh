set -u
export TMPDIR=/tmp
NBX_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --bodies 1048576 --steps 10 --warmup 2 > gpurun_out/r04_bench_rehearsal_1rank_rccl_1m.json 2> gpurun_out/r04_bench_rehearsal_1rank.err
echo rc1=$?
NBX_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --steps 6 --warmup 2 --cpu-baseline none > gpurun_out/r04_bench_rehearsal_2ranks_one_gpu_gloo.json 2> gpurun_out/r04_bench_rehearsal_2ranks.err
echo rc2=$?
NBX_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 4 --steps 6 --warmup 2 --cpu-baseline none > gpurun_out/r04_bench_rehearsal_4ranks_one_gpu_gloo.json 2> gpurun_out/r04_bench_rehearsal_4ranks.err
echo rc3=$?
tail -c 1500 gpurun_out/r04_bench_rehearsal_1rank_rccl_1m.json
