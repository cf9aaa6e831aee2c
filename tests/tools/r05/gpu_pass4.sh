# round 5, GPU pass 4: the whole -m gpu suite, the round's rocprofv3 evidence (profiles/collect.sh), the un-profiled bench lines
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $OUT/t4.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_g.json 2> $OUT/bench_driver_g.err
python bench.py > $OUT/bench_default_g.json 2> $OUT/bench_default_g.err
bash profiles/collect.sh > $OUT/collect.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_h.json 2>> $OUT/bench_driver_g.err
tail -5 $OUT/t4.log
