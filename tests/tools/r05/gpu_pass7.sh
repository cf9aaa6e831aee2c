# round 5, GPU pass 7: bench with the dress rehearsal (three times) against without (twice), and the bench tests (digests: same trajectory)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
Q="--no-cpu-baseline --secondary-legs 0 --large-batch 0 --fence-steps 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 $Q > $OUT/reh1_$i.json 2>> $OUT/reh.err; python bench.py --steps 20 --warmup 5 --rehearsals 0 $Q > $OUT/reh0_$i.json 2>> $OUT/reh.err; done
python bench.py --steps 20 --warmup 5 --rehearsals 2 $Q > $OUT/reh2_1.json 2>> $OUT/reh.err
python -m pytest tests -m gpu -q -k "bench" 2>&1 | tail -8 > $OUT/t7.log
tail -4 $OUT/t7.log
