"""World-size-2 CPU test (gloo) of the multi-GPU plumbing: shard ranges and the episode-return
all-gather.  The step path has no collective; sharding invariance of the kernels themselves is covered
by test_gpu_parity.py::test_full_size_properties_65536."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG


def test_shard_range_partitions():
    from armenv.dist import shard_range
    for total, world in [(524288, 8), (65536, 1), (10, 3), (2, 4), (0, 2)]:
        got = [shard_range(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        sizes = [hi - lo for lo, hi in got]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(524288, 3, 8) == (3 * 65536, 4 * 65536)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from armenv.dist import ReturnGatherer, init_process_group, shard_range
    rank, local_rank, world = init_process_group("gloo", "cpu")
    assert world == 2 and dist.get_backend() == "gloo"
    n_total = 10
    lo, hi = shard_range(n_total, rank, world)
    g = ReturnGatherer(hi - lo, "cpu", world)
    for it in range(3):
        local = torch.arange(lo, hi, dtype=torch.float64) * 10.0 + it        # any float dtype in
        g.launch(local)
        out = g.result()
        want = torch.arange(0, n_total, dtype=torch.float32) * 10.0 + it
        assert torch.equal(out, want), (rank, out, want)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_return_all_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % PKG)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


BARRIER_WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from armenv.dist import ReturnGatherer, ShmBarrier, init_process_group
    rank, local_rank, world = init_process_group("gloo", "cpu")
    b = ShmBarrier(rank, world, timeout_s=60.0)
    # a barrier orders the ranks: nobody leaves barrier k before the slowest rank has entered it
    stamps = np.zeros((50, 2))
    for k in range(50):
        if rank == k %% world:
            time.sleep(0.002)                   # the slow rank of this round
        stamps[k, 0] = time.monotonic()
        b.wait()
        stamps[k, 1] = time.monotonic()
    allst = [None] * world
    dist.all_gather_object(allst, stamps)
    if rank == 0:
        st = np.stack(allst)                    # [world, 50, (enter, leave)]; CLOCK_MONOTONIC is one clock for the node
        assert (st[:, :, 1].min(axis=0) >= st[:, :, 0].max(axis=0) - 1e-6).all()
        assert (st[:, 1:, 0] >= st[:, :-1, 1]).all()
    assert b.epoch == 50
    # the callable form of ReturnGatherer.launch (the producer runs where the gatherer wants it)
    g = ReturnGatherer(3, "cpu", world)
    g.launch(lambda: torch.full((3,), float(rank)))
    assert torch.equal(g.result(), torch.arange(world, dtype=torch.float32).repeat_interleave(3))
    g.order_after_read()                        # no stream on CPU: nothing to order, must not fail
    # the producer that WRITES the send buffer (bench.py: armenv_episode_returns_f32 into the slot), with and without a raw stream argument
    g.warm_up(2)
    g.launch_into(lambda stage: stage.fill_(10.0 + rank))
    assert torch.equal(g.result(), (10.0 + torch.arange(world, dtype=torch.float32)).repeat_interleave(3))
    g.launch_into(lambda stage, stream=None: stage.fill_(20.0 + rank) if stream is None else None, takes_stream=True)
    assert torch.equal(g.result(), (20.0 + torch.arange(world, dtype=torch.float32)).repeat_interleave(3))
    assert g.rccl is None and g.direct_error is None        # gloo: no direct RCCL communicator is attempted
    g.close()
    b.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("world", [2, 4])
def test_shm_barrier_orders_the_ranks_of_a_node(tmp_path, world):
    """armenv.dist.ShmBarrier -- the barrier of bench.py's timed bracket since round 6 (one cache line per rank in a page of
    /dev/shm, named by rank 0 and unlinked once mapped): 50 rounds with a different slow rank each, entry / exit stamps on the
    node's monotonic clock."""
    script = tmp_path / "worker.py"
    script.write_text(BARRIER_WORKER % PKG)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("armenv-barrier-")]


def test_shm_barrier_single_rank_is_the_same_code():
    from armenv.dist import ShmBarrier
    b = ShmBarrier(0, 1)
    for _ in range(1000):
        b.wait()
    assert b.epoch == 1000 and not [f for f in os.listdir("/dev/shm") if f.startswith("armenv-barrier-")]
    b.close()
