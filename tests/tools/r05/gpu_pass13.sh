set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q -k "datd3 or push_full_size or isa or k_loop or contract or n1_push" 2>&1 | tail -12 > $OUT/t13.log
python bench.py --task push --envs-per-gpu 32768 --policy external --steps 600 --no-cpu-baseline --fence-steps 0 > $OUT/bench_push_3.json 2> $OUT/p13.err
tail -5 $OUT/t13.log
