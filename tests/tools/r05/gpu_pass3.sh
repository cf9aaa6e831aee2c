# round 5, GPU pass 3: DATD3 + push tests, the fence reports, the first region against the pre-warm length, the driver's command
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "datd3 or push" 2>&1 | tail -25 > $OUT/t3a.log
python -m pytest tests/test_gpu_fence.py -m gpu -q -s -k "resync or free_running" 2>&1 | grep -E "env-steps|passed|failed|Error|assert" > $OUT/t3_fence.log
for pw in 0 40 150 400; do
  python bench.py --steps 20 --warmup 5 --prewarm-ms $pw --no-cpu-baseline --secondary-legs 0 --large-batch 0 --fence-steps 0 > $OUT/prewarm_$pw.json 2>> $OUT/prewarm.err
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_f.json 2> $OUT/bench_driver_f.err
tail -6 $OUT/t3a.log; tail -3 $OUT/t3_fence.log
