#!/bin/bash
# round 6, pass 4: MFMA / f64 VALU overlap probe (two waves per SIMD); bracket after the gather trims
set -x
mkdir -p gpurun_out/r06
./tests/tools/exp/mfma_valu_overlap_probe > gpurun_out/r06/mfma_valu_overlap_probe.txt 2>&1
cat gpurun_out/r06/mfma_valu_overlap_probe.txt
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/p4_plain_$i.json 2>> gpurun_out/r06/p4_plain.err
  ARMENV_BENCH_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$i bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/p4_rccl_world1_$i.json 2>> gpurun_out/r06/p4_rccl.err
done
python - <<'PY'
import json
for f in ["p4_plain_%d"%i for i in (1,2,3)]+["p4_rccl_world1_%d"%i for i in (1,2,3)]:
    try:
        d=json.loads([l for l in open("gpurun_out/r06/%s.json"%f) if l.startswith('{"metric"')][-1])
        print(f, "value %.3e steps %.3e kernel %.3e bracketed %.3e"%(d["value"],d["value_steps"],d["value_kernel"],d.get("value_bracketed",0)), d["config"].get("collective_transport"), d["config"].get("collective_direct_error"), {k:round(v,1) for k,v in d["config"]["host_us"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --envs-per-gpu 8192 2>&1 | tail -1 | cut -c1-600
