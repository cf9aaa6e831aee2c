set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmcx; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A16="--policy actor_f16x3 --steps 300 --warmup 100 --no-cpu-baseline --fence-steps 0 --large-batch 0 --secondary-legs 0 --repeat-regions 0"
pmc() { local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$OUT/$name" -- python "$REPO/bench.py" "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name '*counter_collection.csv' -exec cp {} "$OUT/${name}_counters.csv" \; ; rm -rf "$OUT/$name"; }
pmc a1 SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS -- $A16
pmc a2 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $A16
pmc a3 SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS -- $A16
pmc a4 SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_IFETCH -- $A16
pmc a5 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_CVT -- $A16
ls -la $OUT
