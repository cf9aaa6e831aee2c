# round 5, GPU pass 12: the whole -m gpu suite on the final push defaults + bench lines (driver twice, default once)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/t12.log
python -m pytest tests/test_gpu_fence.py -m gpu -q -s -k "resync or free_running" 2>&1 | grep -E "env-steps|passed|failed" > $OUT/t12_fence.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_j1.json 2> $OUT/bench_driver_j.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_j2.json 2>> $OUT/bench_driver_j.err
python bench.py > $OUT/bench_default_j.json 2>> $OUT/bench_driver_j.err
tail -5 $OUT/t12.log
