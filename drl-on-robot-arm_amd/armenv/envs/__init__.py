"""Env registry with the reference's names (/root/reference/envs/__init__.py:1-4), so that
``getattr(envs, opt.env)`` (/root/reference/main.py:83) resolves here too."""
from .rl_reach_env import RLReachEnv
from .rl_push_env import RLPushEnv
from .rl_pick_env import RLPickEnv
from .batched import BatchedArmEnv, BatchedReachEnv, BatchedPushEnv, BatchedPickEnv, diana_cam_reach_kinematics
from .pipelined import PipelinedEnv

__all__ = ["RLReachEnv", "RLPushEnv", "RLPickEnv", "BatchedArmEnv", "BatchedReachEnv", "BatchedPushEnv", "BatchedPickEnv", "diana_cam_reach_kinematics", "PipelinedEnv"]
