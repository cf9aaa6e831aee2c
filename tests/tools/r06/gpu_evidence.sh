#!/bin/bash
# round 6, the evidence pass on the final tree (through gpurun): the whole -m gpu suite with durations, smoke, the fence report lines,
# the driver's line (twice) and the default line un-profiled, the one-rank RCCL line beside a plain one, the overlap probe, a DADDPG run
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -25 > $OUT/gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1
python -m pytest tests/test_gpu_fence.py -m gpu -q -s -k "resync or free_running" 2>&1 | grep -E "env-steps|passed|failed" > $OUT/fence_final.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_1.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_2.json 2>> $OUT/bench.err
python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err
A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0"
python bench.py $A > $OUT/bench_plain_final.json 2>> $OUT/bench.err
ARMENV_BENCH_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py $A > $OUT/bench_rccl_world1_final.json 2>> $OUT/bench.err
ARMENV_DIST_DIRECT_RCCL=0 ARMENV_BENCH_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py $A > $OUT/bench_rccl_world1_torchdist_final.json 2>> $OUT/bench.err
python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_8ranks_one_gpu.json 2>> $OUT/bench.err
./tests/tools/exp/mfma_valu_overlap_probe > $OUT/mfma_valu_overlap_probe.txt 2>&1
python -m armenv.train --iterations 200 --algo daddpg > $OUT/train_reach_daddpg.jsonl 2>> $OUT/bench.err
python -m armenv.train --iterations 200 > $OUT/train_reach_td3.jsonl 2>> $OUT/bench.err
tail -4 $OUT/gputest_final.log; tail -1 $OUT/smoke_final.log; tail -2 $OUT/train_reach_daddpg.jsonl; tail -1 $OUT/train_reach_td3.jsonl
