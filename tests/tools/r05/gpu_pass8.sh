set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
Q="--no-cpu-baseline --secondary-legs 0 --large-batch 0 --fence-steps 0 --ab-regions 0"
for i in 1 2; do python bench.py --steps 20 --warmup 5 --repeat-same-rows $Q > $OUT/same_$i.json 2>> $OUT/same.err; done
python bench.py --steps 20 --warmup 5 $Q > $OUT/fresh_1.json 2>> $OUT/same.err
