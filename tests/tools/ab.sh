#!/bin/bash
# A/B timing of two builds of libarmenv.so inside ONE GPU session (box-to-box variance is +-2 %):
#   cp drl-on-robot-arm_amd/armenv/libarmenv.so drl-on-robot-arm_amd/build/libarmenv_prev.so   (before the change)
#   gpurun -- 'bash tests/tools/ab.sh [bench args]'
# alternates the two libraries three times and prints env-steps/s and us per step of each run.
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"]*1e3,3), (d.get("step_api") or {}).get("avg_launch_us"))'
for r in 1 2 3; do
  ARMENV_LIB=$PWD/drl-on-robot-arm_amd/build/libarmenv_prev.so python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$P" prev
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$P" new
done
