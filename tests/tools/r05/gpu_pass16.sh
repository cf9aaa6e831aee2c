# A/B in one session: in-kernel random policy with / without the speculative draws, external actions beside them
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
Q="--steps 2000 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0 --repeat-regions 0"
for i in 1 2 3; do
  python bench.py --policy random $Q > $OUT/ab_spec_$i.json 2>> $OUT/p16.err
  ARMENV_LIB=$REPO/tests/tools/exp/libarmenv_nospec.so python bench.py --policy random $Q > $OUT/ab_nospec_$i.json 2>> $OUT/p16.err
  python bench.py --policy external $Q > $OUT/ab_ext_$i.json 2>> $OUT/p16.err
done
