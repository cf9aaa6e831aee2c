# round 5, GPU pass 19 (final tree): the whole -m gpu suite, smoke, the fence report lines, pick / push / driver bench lines
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/t19.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke19.log 2>&1
python -m pytest tests/test_gpu_fence.py -m gpu -q -s -k "resync or free_running" 2>&1 | grep -E "env-steps|passed|failed" > $OUT/t19_fence.log
python bench.py --task pick --envs-per-gpu 32768 --steps 1000 --no-cpu-baseline --fence-steps 0 > $OUT/bench_pick_2.json 2> $OUT/p19.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_m1.json 2>> $OUT/p19.err
tail -5 $OUT/t19.log; tail -1 $OUT/smoke19.log
