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
