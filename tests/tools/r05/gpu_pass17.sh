set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q -k "pick or cube or half_filled or datd3 or n1_cube or training" 2>&1 | tail -25 > $OUT/t17.log
python bench.py --task pick --envs-per-gpu 32768 --steps 1000 --no-cpu-baseline --fence-steps 0 > $OUT/bench_pick_1.json 2> $OUT/p17.err
tail -8 $OUT/t17.log
