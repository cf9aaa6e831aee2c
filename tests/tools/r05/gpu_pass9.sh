set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
Q="--no-cpu-baseline --secondary-legs 0 --large-batch 0 --fence-steps 0 --ab-regions 0"
for i in 1 2; do ARMENV_BENCH_COUNT_REPEATS=1 python bench.py --steps 20 --warmup 5 $Q > $OUT/cnt_$i.json 2>> $OUT/cnt.err; done
