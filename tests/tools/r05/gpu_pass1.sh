# round 5, GPU pass 1: contract + launcher tests, the driver's command twice (clock probes, restore regimes), the f16x3 actor's
# finer phase stamps and the ring protocol 2 vs 3 positions ahead (timeline builds, A/B of the product builds, goldens on the variant)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "plain_python or bench_line_contract or ipelined or driver_shape" 2>&1 | tail -15 > $OUT/t1.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_c.json 2> $OUT/bench_driver_c.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_d.json 2>> $OUT/bench_driver_c.err
A16="--policy actor_f16x3 --steps 1000 --warmup 200 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0 --repeat-regions 0"
python tests/tools/exp/run_actor_timeline.py > $OUT/actor_timeline_ahead2.txt 2>&1
ARMENV_TL_LIB=$REPO/tests/tools/exp/libarmenv_tl_ahead3.so python tests/tools/exp/run_actor_timeline.py > $OUT/actor_timeline_ahead3.txt 2>&1
for i in 1 2 3; do
  python bench.py $A16 > $OUT/actor_ab_ahead2_$i.json 2>> $OUT/actor_ab.err
  ARMENV_LIB=$REPO/tests/tools/exp/libarmenv_ahead3.so python bench.py $A16 > $OUT/actor_ab_ahead3_$i.json 2>> $OUT/actor_ab.err
done
ARMENV_LIB=$REPO/tests/tools/exp/libarmenv_ahead3.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "actor or config2" 2>&1 | tail -8 > $OUT/t_ahead3.log
cat $OUT/t1.log $OUT/t_ahead3.log
