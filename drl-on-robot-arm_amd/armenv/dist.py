"""Multi-GPU plumbing: one process per GPU, envs sharded by contiguous global index ranges, and the one
collective the path has -- an all-gather of per-env episode returns for logging
(what /root/reference/main.py:130,150 plots from a single env).  The step path itself exchanges nothing.

Works with backend "nccl" (= RCCL over xGMI on ROCm) on GPUs and with "gloo" on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    """RANK / LOCAL_RANK / WORLD_SIZE as torch.distributed.run exports them (1-process default)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(total_envs, rank, world):
    """Contiguous [lo, hi) of global env indices owned by `rank`; the first (total % world) ranks own one
    more.  `lo` is the handle's env_id_offset, so goals/episodes do not depend on the sharding."""
    if not (0 <= rank < world) or total_envs < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend=None, device=None):
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = torch.device(device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


class ReturnGatherer:
    """All-gathers a per-env f32 vector (episode returns) to every rank, off the step's critical path:
    on GPUs the collective runs on a side stream that waits for the producer stream only."""

    def __init__(self, n_local, device, world=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.device = torch.device(device)
        self.n_local = n_local
        self.out = torch.zeros(self.world * n_local, dtype=torch.float32, device=self.device)
        self.stage = torch.zeros(n_local, dtype=torch.float32, device=self.device)
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._work = None

    def launch(self, local_returns):
        """Enqueue the gather of `local_returns` ([n_local], any float dtype).  Non-blocking on GPUs."""
        if self.side is not None and self.world > 1 and dist.get_backend() != "nccl":
            # debugging path (several ranks sharing one GPU cannot use RCCL): stage through the host with gloo
            host = local_returns.detach().to("cpu", torch.float32)
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host)
            self.out.copy_(torch.cat(parts))
            return
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side):
                self.stage.copy_(local_returns)
                if self.world > 1:
                    self._work = dist.all_gather_into_tensor(self.out, self.stage, async_op=True)
                else:
                    self.out.copy_(self.stage)
        else:
            self.stage.copy_(local_returns)
            if self.world > 1:
                parts = [torch.empty_like(self.stage) for _ in range(self.world)]
                dist.all_gather(parts, self.stage)
                self.out.copy_(torch.cat(parts))
            else:
                self.out.copy_(self.stage)

    def result(self):
        """Block until the last launched gather is complete; returns the [world * n_local] tensor."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return self.out
