#!/bin/bash
# Collects a round's rocprofv3 evidence on the MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh'        (or `bash profiles/collect.sh pmc` for the counter passes only,
#                                                               `bash profiles/collect.sh pick` to redo the pick row alone)
# Writes under gpurun_out/prof/; profiles/aggregate.py turns the outputs into the summaries kept in profiles/ (r06_*).
# PMC passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
MODE=${1:-all}
[ "$MODE" = "all" ] && rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, bench args...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python "$REPO/bench.py" "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name '*kernel_stats.csv' -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  find "$OUT/$name" -name '*kernel_trace.csv' -exec cp {} "$OUT/${name}_kernel_trace.csv" \;    # per dispatch: aggregate.py splits a symbol's launches by their length
  grep -h "^{\"metric\"" "$OUT/$name.log" | tail -1 > "$OUT/${name}_bench.json"
}
if [ "$MODE" = "all" ]; then
stats driver --steps 20 --warmup 5               # the driver's own command (with cpu_baseline and the parity-fence leg)
stats default                                     # bench.py defaults: 5 000 steps, 100 per launch
stats actor_f16x3 --policy actor_f16x3 --steps 1000 --no-cpu-baseline --fence-steps 0
stats actor_f32 --policy actor --steps 500 --no-cpu-baseline --fence-steps 0
stats push32768 --task push --envs-per-gpu 32768 --steps 1000 --no-cpu-baseline
stats pick32768 --task pick --envs-per-gpu 32768 --steps 1000 --no-cpu-baseline
stats f32engine --precision 32 --steps 1000 --no-cpu-baseline --fence-steps 0 --secondary-legs 0
fi
if [ "$MODE" = "driver" ]; then
stats driver --steps 20 --warmup 5
stats driver2 --steps 20 --warmup 5
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
exit 0
fi
if [ "$MODE" = "pick" ]; then
stats pick32768 --task pick --envs-per-gpu 32768 --steps 1000 --no-cpu-baseline
fi
# one small counter set per pass: a set the hardware cannot collect in one pass makes rocprofv3 abort and then hang in its
# signal handler (FETCH_SIZE + WRITE_SIZE + GRBM_GUI_ACTIVE did), hence the timeouts
pmc() {   # name, counters..., then -- bench args
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$OUT/$name" -- python "$REPO/bench.py" "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name '*counter_collection.csv' -exec cp {} "$OUT/${name}_counters.csv" \;
}
if [ "$MODE" = "pick" ]; then
pmc pmc_pick SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- --task pick --envs-per-gpu 32768 --steps 500 --warmup 50 --no-cpu-baseline --fence-steps 0
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la "$OUT"
exit 0
fi
B100="--steps 500 --warmup 50 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"
B20="--steps 200 --warmup 20 --rollout-steps 20 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"    # the driver's launch shape, ten launches
if [ "$MODE" = "all" ]; then
pmc pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -- $B100
pmc pmc2 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- $B100
pmc pmc5 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $B100
fi
pmc pmc3 FETCH_SIZE -- $B100
pmc pmc4 WRITE_SIZE -- $B100
pmc pmc3_T20 FETCH_SIZE -- $B20
pmc pmc4_T20 WRITE_SIZE -- $B20
# the launch shapes of bench.py's config3 / config4 legs (100 steps per launch): HBM traffic for their `roofline.traffic`
A32="--policy actor --steps 300 --warmup 100 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"
A16="--policy actor_f16x3 --steps 500 --warmup 100 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"
PU="--task push --envs-per-gpu 32768 --steps 500 --warmup 100 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"
pmc pmc3_actor FETCH_SIZE -- $A32
pmc pmc4_actor WRITE_SIZE -- $A32
pmc pmc3_actor_f16x3 FETCH_SIZE -- $A16
pmc pmc4_actor_f16x3 WRITE_SIZE -- $A16
pmc pmc3_push FETCH_SIZE -- $PU
pmc pmc4_push WRITE_SIZE -- $PU
# the fused two-actor policies (bench.py's datd3_fused / daddpg_fused legs)
D3="--policy datd3 --steps 300 --warmup 100 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"
DD="--policy daddpg --steps 300 --warmup 100 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0"
pmc pmc3_datd3 FETCH_SIZE -- $D3
pmc pmc4_datd3 WRITE_SIZE -- $D3
pmc pmc3_daddpg FETCH_SIZE -- $DD
pmc pmc4_daddpg WRITE_SIZE -- $DD
if [ "$MODE" = "all" ]; then
pmc pmc_actor SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- --policy actor_f16x3 --steps 200 --no-cpu-baseline --fence-steps 0
pmc pmc_actor3 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES -- --policy actor_f16x3 --steps 200 --no-cpu-baseline --fence-steps 0
pmc pmc_actor2 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- --policy actor_f16x3 --steps 200 --no-cpu-baseline --fence-steps 0
pmc pmc_push SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- --task push --envs-per-gpu 32768 --steps 500 --warmup 50 --no-cpu-baseline --fence-steps 0
pmc pmc_pick SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- --task pick --envs-per-gpu 32768 --steps 500 --warmup 50 --no-cpu-baseline --fence-steps 0
fi
# keep only the summaries (the raw trees are large)
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la "$OUT"
