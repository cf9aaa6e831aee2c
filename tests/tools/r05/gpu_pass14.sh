set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q -k "random_policy or fused_policy or half_filled or lockstep or waves or soak or rollout_equals" 2>&1 | tail -6 > $OUT/t14.log
for i in 1 2; do python bench.py --policy random --steps 2000 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0 --repeat-regions 0 > $OUT/inkernel_spec_$i.json 2>> $OUT/p14.err; done
tail -3 $OUT/t14.log
