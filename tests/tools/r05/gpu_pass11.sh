set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
Q="--no-cpu-baseline --secondary-legs 0 --large-batch 0 --fence-steps 0 --ab-regions 0"
for d in 0.8 0.5; do for i in 1 2; do ARMENV_BENCH_PREWARM_DUTY=$d python bench.py --steps 20 --warmup 5 $Q > $OUT/duty_${d}_$i.json 2>> $OUT/duty.err; done; done
for i in 1 2; do python bench.py --steps 20 --warmup 5 --busy-ahead-ms 3 $Q > $OUT/ahead3_$i.json 2>> $OUT/duty.err; done
