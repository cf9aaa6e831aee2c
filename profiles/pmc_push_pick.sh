set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_pp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in push pick; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/$t -- python $REPO/bench.py --task $t --envs-per-gpu 32768 --steps 500 --warmup 100 --no-cpu-baseline > $OUT/$t.log 2>&1
  find $OUT/$t -name '*counter_collection.csv' -exec cp {} $OUT/${t}_counters.csv \;
  grep -h '^{"metric"' $OUT/$t.log | tail -1 > $OUT/${t}_bench.json
  rm -rf $OUT/$t
done
ls -la $OUT
