# round 5, GPU pass 5: push / DATD3 / recorded-run tests on the final kernels, then the legs' timings twice
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "datd3 or push or cube or n1_ or random_policy or fused_policy" 2>&1 | tail -12 > $OUT/t5.log
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_i$i.json 2>> $OUT/bench_driver_i.err
  python bench.py --task push --envs-per-gpu 32768 --steps 1000 --no-cpu-baseline --fence-steps 0 > $OUT/bench_push_$i.json 2>> $OUT/bench_driver_i.err
done
tail -4 $OUT/t5.log
