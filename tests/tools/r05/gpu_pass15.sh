# round 5, GPU pass 15 (final tree): the whole -m gpu suite, the rocprofv3 evidence again, un-profiled bench lines
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/t15.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke15.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_k1.json 2> $OUT/bench_driver_k.err
python bench.py > $OUT/bench_default_k.json 2>> $OUT/bench_driver_k.err
bash profiles/collect.sh > $OUT/collect15.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_k2.json 2>> $OUT/bench_driver_k.err
tail -5 $OUT/t15.log; tail -1 $OUT/smoke15.log
