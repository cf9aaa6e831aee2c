# round 5, GPU pass 2: the whole -m gpu suite on the ABI-5 library, then the driver's command
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $OUT/t2.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_e.json 2> $OUT/bench_driver_e.err
tail -5 $OUT/t2.log
