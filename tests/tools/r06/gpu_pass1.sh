#!/bin/bash
# round 6, pass 1: the new bracket -- plain N=1 vs one rank down the RCCL path, same session; the bench tests
set -x
mkdir -p gpurun_out/r06
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_a.json 2> gpurun_out/r06/bench_driver_a.err
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/bench_plain_$i.json 2>> gpurun_out/r06/bench_plain.err
  ARMENV_BENCH_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$i bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary-legs 0 --fence-steps 0 --large-batch 0 --ab-regions 0 > gpurun_out/r06/bench_rccl_world1_$i.json 2>> gpurun_out/r06/bench_rccl.err
done
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_2ranks_1gpu.json 2> gpurun_out/r06/bench_2ranks_1gpu.err
python -m pytest tests -m gpu -x -q -k "bench" 2>&1 | tail -15 > gpurun_out/r06/pytest_bench.txt
cat gpurun_out/r06/pytest_bench.txt
python - <<'PY'
import json
for f in ["bench_driver_a"]+["bench_plain_%d"%i for i in (1,2,3)]+["bench_rccl_world1_%d"%i for i in (1,2,3)]+["bench_2ranks_1gpu"]:
    try:
        d=json.loads([l for l in open("gpurun_out/r06/%s.json"%f) if l.startswith('{"metric"')][-1])
        print(f, "value %.3e steps %.3e kernel %.3e bracketed %.3e med %s"%(d["value"],d["value_steps"],d["value_kernel"],d.get("value_bracketed",0),d.get("value_median")), {k:round(v,1) for k,v in d["config"]["host_us"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
