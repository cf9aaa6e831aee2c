#!/bin/bash
# A/B timing of two builds of libarmenv.so with the kernel-time probe (time_rollout.py) inside ONE GPU session:
#   gpurun -- 'bash tests/tools/ab_time.sh drl-on-robot-arm_amd/build/ab/libarmenv_x.so [rounds] -- <time_rollout args>'
# alternates `other` and the current library `rounds` times (default 3).
OTHER=$PWD/$1; shift
R=3; if [ "$1" != "--" ]; then R=$1; shift; fi; shift
for r in $(seq $R); do
  ARMENV_LIB=$OTHER python tests/tools/time_rollout.py "$@" 2>&1 | grep -v amdgpu.ids
  python tests/tools/time_rollout.py "$@" 2>&1 | grep -v amdgpu.ids
done
