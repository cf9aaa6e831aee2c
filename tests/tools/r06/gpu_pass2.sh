#!/bin/bash
# round 6, pass 2: direct-RCCL gather + bracket; DADDPG / DARC; N=1 cube rewards; full gpu suite
set -x
mkdir -p gpurun_out/r06
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/p2_plain_$i.json 2>> gpurun_out/r06/p2_plain.err
  ARMENV_BENCH_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$i bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/p2_rccl_world1_$i.json 2>> gpurun_out/r06/p2_rccl.err
done
ARMENV_DIST_DIRECT_RCCL=0 ARMENV_BENCH_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29507 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/p2_rccl_world1_torchdist.json 2>> gpurun_out/r06/p2_rccl.err
python - <<'PY'
import json
for f in ["p2_plain_%d"%i for i in (1,2,3)]+["p2_rccl_world1_%d"%i for i in (1,2,3)]+["p2_rccl_world1_torchdist"]:
    try:
        d=json.loads([l for l in open("gpurun_out/r06/%s.json"%f) if l.startswith('{"metric"')][-1])
        print(f, "value %.3e steps %.3e kernel %.3e bracketed %.3e"%(d["value"],d["value_steps"],d["value_kernel"],d.get("value_bracketed",0)), d["config"].get("collective_transport"), d["config"].get("collective_direct_error"), {k:round(v,1) for k,v in d["config"]["host_us"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/r06/p2_rccl.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06/pytest_gpu_p2.txt
cat gpurun_out/r06/pytest_gpu_p2.txt
