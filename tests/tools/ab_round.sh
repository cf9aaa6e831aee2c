#!/bin/bash
# A/B of this tree against another checkout of the repo (e.g. the previous round's, `git worktree add
# drl-on-robot-arm_amd/build/r02tree <commit>` + make there) inside ONE GPU session: each side runs its own time_rollout.py
# with its own library and Python package, alternating `rounds` times.
#   gpurun -- 'bash tests/tools/ab_round.sh drl-on-robot-arm_amd/build/r02tree 3 -- --task reach --pre 600'
OTHER=$PWD/$1; shift
R=$1; shift; shift
for r in $(seq $R); do
  echo -n "other   "; python $OTHER/tests/tools/time_rollout.py "$@" 2>&1 | grep -v amdgpu.ids
  echo -n "current "; python tests/tools/time_rollout.py "$@" 2>&1 | grep -v amdgpu.ids
done
