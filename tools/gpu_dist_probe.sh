#!/bin/bash
# what a process group costs the default line: none, RCCL set up after the model's streams (bench.py's order), RCCL
# set up first (VITS_BENCH_DIST_EARLY=1; the generator then stays on one stream), the same with two generator streams
# forced, and under torch.distributed.run as the driver starts it
out=gpurun_out/dist_probe; mkdir -p $out; log=$out/log.txt; : > $log; : > $out/err.txt
B="--steps 20 --warmup 5 --no-sub-results --no-latency --no-cpu-baseline"
R="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 VITS_BENCH_FORCE_DIST=1"
run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $log; }
echo no_group >> $log; timeout -k 10 200 python3 bench.py $B 2>>$out/err.txt | run || exit 1
echo rccl_after_the_streams >> $log; env $R timeout -k 10 200 python3 bench.py $B 2>>$out/err.txt | run || exit 1
echo rccl_first >> $log; env $R VITS_BENCH_DIST_EARLY=1 timeout -k 10 200 python3 bench.py $B 2>>$out/err.txt | run || exit 1
echo rccl_first_two_generator_streams >> $log; env $R VITS_BENCH_DIST_EARLY=1 VITS_BATCHES_GEN_STREAMS=2 timeout -k 10 200 python3 bench.py $B 2>>$out/err.txt | run || exit 1
echo torchrun_one_rank >> $log; VITS_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py $B 2>>$out/err.txt | run || exit 1
echo torchrun_one_rank_c4 >> $log; VITS_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29514 bench.py --workload c4 --utterances 2000 --no-cpu-baseline 2>>$out/err.txt | run || exit 1
echo c4_no_group >> $log; timeout -k 10 300 python3 bench.py --workload c4 --utterances 2000 --no-cpu-baseline 2>>$out/err.txt | run || exit 1
cat $log
