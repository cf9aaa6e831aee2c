"""Multi-GPU plumbing: one process per GPU, envs sharded by contiguous global index ranges, and the one
collective the path has -- an all-gather of per-env episode returns for logging
(what /root/reference/main.py:130,150 plots from a single env).  The step path itself exchanges nothing.

Works with backend "nccl" (= RCCL over xGMI on ROCm) on GPUs and with "gloo" on CPU tensors (tests).
"""
import ctypes
import mmap
import os
import threading
import time
import uuid

import numpy as np
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


def init_process_group(backend=None, device=None, force=False):
    """force: initialise the group for a one-rank job as well (RCCL accepts a communicator of one rank: a 1-GPU box then runs
    the same collective calls an 8-GPU node does; needs MASTER_ADDR / MASTER_PORT as torch.distributed.run exports them)."""
    rank, local_rank, world = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = torch.device(device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


class ShmBarrier:
    """Barrier of the ranks of ONE node through a page of shared memory: every rank owns one cache line and stores the number of
    the barrier it has reached there; wait() returns when every line holds at least that number (single writer per line, aligned
    64-bit stores: no atomics, nothing to reset, no sense flag to flip).  The ranks of this build are local by construction (one
    process per GPU of a node), so the barrier that brackets a timed region costs a microsecond or two instead of a collective's
    launch + completion (`dist.barrier()` on RCCL is an all-reduce kernel: tens of microseconds, and it puts the launch stream
    to work).  The file under /dev/shm is unlinked as soon as every rank has mapped it: nothing is left behind by a crash.
    With one rank and no process group it is the same code over one line."""

    LINE = 8        # int64 slots per rank (64 bytes)

    def __init__(self, rank=None, world=None, timeout_s=300.0):
        r, _, w = env_rank_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.timeout_s = float(timeout_s)
        grouped = dist.is_initialized() and dist.get_world_size() > 1
        name = ["armenv-barrier-%d-%s" % (os.getpid(), uuid.uuid4().hex[:12])]
        if grouped:       # rank 0 names the page
            dist.broadcast_object_list(name, src=0)
        path = os.path.join("/dev/shm", name[0])
        size = 8 * self.LINE * self.world
        fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
        try:
            os.ftruncate(fd, size)          # every rank asks for the same size: the page's contents are never cut
            self._mm = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self._slots = np.frombuffer(self._mm, dtype=np.int64)[::self.LINE]
        self.epoch = 0
        if grouped:
            dist.barrier()                  # everyone has mapped the page ...
        if self.rank == 0:
            try:
                os.unlink(path)             # ... so its name can go (the mappings stay)
            except FileNotFoundError:
                pass

    def wait(self):
        self.epoch += 1
        s, e = self._slots, self.epoch
        s[self.rank] = e
        spins = 0
        while (s < e).any():
            spins += 1
            if spins & 0xFFFF == 0:
                if spins == 0x10000:
                    t0 = time.monotonic()
                elif time.monotonic() - t0 > self.timeout_s:
                    raise TimeoutError("ShmBarrier: a rank did not arrive within %.0f s (slots %s, waiting for %d)"
                                       % (self.timeout_s, s.tolist(), e))

    def close(self):
        self._slots = None
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]       # rccl.h: NCCL_UNIQUE_ID_BYTES (c_ubyte: a c_char array reads back cut at its first NUL)


class RcclComm:
    """A communicator of this job's ranks made with RCCL's own C API, through the librccl.so this process has already loaded (the one
    torch.distributed's "nccl" backend runs on): ncclGetUniqueId on rank 0, the id handed round with the job's process group, then
    ncclCommInitRank on every rank.  Why, beside torch.distributed: the one collective of this path -- the logging all-gather -- is
    issued from inside the rollout loop, and its ISSUE is what the loop pays; `dist.all_gather_into_tensor(async_op=True)` costs the
    host 100-250 us per call (profiles/r06_bench_rccl_world1_torchdist.json: host_us.gather_issue 125), as long as a 20-step rollout
    launch runs; ncclAllGather on a stream the caller names is one C call (profiles/r06_bench_rccl_world1.json: 42 us for the whole issue).  Creation happens in a helper thread with a time limit: a
    rank that cannot make the communicator reports so, the ranks agree (all-reduce), and the caller falls back to torch.distributed."""

    FLOAT32 = 7           # rccl.h ncclDataType_t

    def __init__(self, device, timeout_s=120.0):
        self.comm = None
        self.lib = None
        rank, world = dist.get_rank(), dist.get_world_size()
        ok, why = 1, ""
        try:
            path = next((l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l or "libnccl" in l), None)
            if path is None:
                raise OSError("no RCCL library is loaded in this process")
            lib = ctypes.CDLL(path)
            lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = ctypes.c_int, [ctypes.POINTER(_NcclUniqueId)]
            lib.ncclCommInitRank.restype = ctypes.c_int
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
            lib.ncclAllGather.restype = ctypes.c_int
            lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = ctypes.c_int, [ctypes.c_void_p]
            lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = ctypes.c_char_p, [ctypes.c_int]
            self.lib = lib
        except Exception as e:      # noqa: BLE001
            ok, why = 0, "binding: %s" % e
        uid = _NcclUniqueId()
        if ok and rank == 0 and self.lib.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            ok, why = 0, "ncclGetUniqueId failed"
        box = [ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if ok:
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
            res = {}

            def make():
                torch.cuda.set_device(device)
                h = ctypes.c_void_p()
                res["rc"] = self.lib.ncclCommInitRank(ctypes.byref(h), world, uid, rank)
                res["h"] = h
            th = threading.Thread(target=make, daemon=True)
            th.start()
            th.join(timeout_s)
            if th.is_alive():
                ok, why = 0, "ncclCommInitRank did not return within %.0f s" % timeout_s
            elif res.get("rc", -1) != 0:
                ok, why = 0, "ncclCommInitRank: %s" % self.lib.ncclGetErrorString(res["rc"]).decode()
            else:
                self.comm = res["h"]
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if self.comm is not None:
                self.lib.ncclCommDestroy(self.comm)
                self.comm = None
            raise RuntimeError("RcclComm: not every rank could create the communicator (this rank: %s)" % (why or "ok"))
        self.world = world

    def all_gather_f32(self, send_ptr, recv_ptr, count, stream_ptr):
        rc = self.lib.ncclAllGather(send_ptr, recv_ptr, count, self.FLOAT32, self.comm, stream_ptr)
        if rc != 0:
            raise RuntimeError("ncclAllGather: %s" % self.lib.ncclGetErrorString(rc).decode())

    def close(self):
        if self.comm is not None:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


class ReturnGatherer:
    """All-gathers a per-env f32 vector (episode returns) to every rank, off the step's critical path:
    on GPUs the collective runs on a side stream that waits for the producer stream only.

    Two (stage, out) slots alternate between launches, and a slot is not refilled before the collective that last used it
    has completed: an asynchronous all-gather may still be reading `stage` / writing `out` when the next launch arrives
    (round 1 had one slot and refilled it unconditionally -- a write-after-read hazard).  all_gather_into_tensor needs the
    same n_local on every rank: `shard_range` gives uneven shards when world does not divide the env count, so the
    constructor checks."""

    def __init__(self, n_local, device, world=None, collective=None, direct=None):
        """direct: issue the collective with RCCL's C API on the side stream (RcclComm) instead of torch.distributed's call -- default
        on GPUs with the "nccl" backend unless ARMENV_DIST_DIRECT_RCCL=0; falls back to torch.distributed (with `direct_error` set)
        when the communicator cannot be made."""
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        # collective: go through torch.distributed even when there is only one rank (a one-rank RCCL communicator: what a
        # 1-GPU box can execute of the multi-GPU path); default: only when there is someone to exchange with
        self.collective = (self.world > 1) if collective is None else bool(collective)
        if self.collective and not dist.is_initialized():
            raise RuntimeError("ReturnGatherer(collective=True) needs an initialised process group")
        self.device = torch.device(device)
        self.n_local = int(n_local)
        self.slots = [dict(stage=torch.zeros(self.n_local, dtype=torch.float32, device=self.device),
                           out=torch.zeros(self.world * self.n_local, dtype=torch.float32, device=self.device), work=None)
                      for _ in range(2)]
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.launches = 0
        self._last = None
        self.read_done = None        # (GPU) event on the side stream: the last launch's source has been copied into its slot
        self._read_evs = [torch.cuda.Event(), torch.cuda.Event()] if self.side is not None else None    # re-used: one per slot
        self._fork_ev = torch.cuda.Event() if self.side is not None else None
        self.rccl, self.direct_error = None, None
        if direct is None:
            direct = os.environ.get("ARMENV_DIST_DIRECT_RCCL", "1") != "0"
        if direct and self.collective and self.side is not None and dist.is_initialized() and dist.get_backend() == "nccl":
            try:
                self.rccl = RcclComm(self.device)
            except Exception as e:      # noqa: BLE001
                self.direct_error = str(e)
        if self.collective and dist.is_initialized():
            host = dist.get_backend() != "nccl"
            t = torch.tensor([self.n_local, -self.n_local], dtype=torch.int64, device="cpu" if host else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if int(t[0]) != -int(t[1]):
                raise ValueError("ReturnGatherer: every rank must own the same number of envs (pad the last shard)")

    def _host_backend(self):
        return self.side is not None and self.collective and dist.get_backend() != "nccl"

    def _check(self, t):
        if tuple(t.shape) != (self.n_local,):
            raise ValueError(f"local_returns must have shape ({self.n_local},)")
        return t

    _fills = False      # launch_into: the producer writes the f32 vector straight into the slot's send buffer
    _raw_stream_ok = False

    def launch_into(self, fill, takes_stream=False):
        """As launch(callable), for a producer that WRITES the vector: `fill(stage)` is called with the slot's f32 send buffer
        [n_local] (on GPUs with the side stream current) -- BatchedArmEnv.episode_returns_f32(out=stage): one kernel, no
        intermediate tensor, no conversion copy.  takes_stream: the producer is `fill(stage, stream)` and enqueues on the raw HIP
        stream it is handed (the direct-RCCL path then enters no torch stream context at all)."""
        self._fills, self._raw_stream_ok = True, bool(takes_stream)
        try:
            self.launch((lambda st, s=None: fill(st, s)) if takes_stream else fill)
        finally:
            self._fills = self._raw_stream_ok = False

    def warm_up(self, rounds=3):
        """A few gathers of zeros, waited for: the first collectives on a fresh communicator set up channels and load kernels
        (the first ncclAllGather of a job cost the issuing host 214 us against 40-68 afterwards);
        a rollout loop that logs from its first step on does not want that inside it."""
        for _ in range(int(rounds)):
            self.launch_into(lambda st: st.zero_())
            self.result()
        if self.side is not None:
            torch.cuda.current_stream(self.device).synchronize()

    def launch(self, local_returns):
        """Enqueue the gather of `local_returns` ([n_local], any float dtype).  Non-blocking on GPUs: the side stream waits for
        the work queued on the producer stream so far (the tensor's producer, and every consumer of an earlier result()).

        `local_returns` may be a callable returning that tensor: on GPUs it is called with the SIDE stream current, so the kernels
        that produce the vector (armenv_episode_stats) are enqueued there too and the producer stream carries nothing of the
        logging path -- its next synchronise closes on the env steps alone.  Whoever then overwrites what the callable read (the
        next env launch) orders itself behind `read_done` (`order_after_read()`); `result()` does so as well."""
        producer = local_returns if callable(local_returns) else None
        if producer is None:
            self._check(local_returns)
        slot = self.slots[self.launches & 1]
        self.launches += 1
        self._last = slot
        prev, slot["work"] = slot["work"], None      # the collective that used this slot two launches ago
        if self._host_backend():
            # debugging path (several ranks sharing one GPU cannot use RCCL): stage through the host with gloo
            if producer is not None:
                local_returns = self._produce(producer, slot)
            host = local_returns.detach().to("cpu", torch.float32)
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host)
            slot["out"].copy_(torch.cat(parts))
            return
        if self.rccl is not None:
            # RCCL's own call on the side stream: everything of this slot's earlier use is in that stream's order already
            cur = torch.cuda.current_stream(self.device)
            self._fork_ev.record(cur)          # the producer of the vector; consumers of the result() of two launches ago
            self.side.wait_event(self._fork_ev)
            if producer is not None and self._fills and self._raw_stream_ok:
                producer(slot["stage"], self.side.cuda_stream)     # one C call on the side stream, no stream context to enter
            else:
                with torch.cuda.stream(self.side):
                    src = local_returns if producer is None else self._produce(producer, slot)
                    if src is not slot["stage"]:
                        slot["stage"].copy_(src)
                    if producer is None:
                        local_returns.record_stream(self.side)
            self.read_done = self._read_evs[(self.launches - 1) & 1]
            self.read_done.record(self.side)
            self.rccl.all_gather_f32(slot["stage"].data_ptr(), slot["out"].data_ptr(), self.n_local, self.side.cuda_stream)
            return
        if self.side is not None:
            cur = torch.cuda.current_stream(self.device)
            # also orders the refill of this slot's `out` behind consumer kernels that still read the tensor result()
            # returned two launches ago (they were enqueued on `cur` before this call)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                if prev is not None:
                    prev.wait()               # orders the SIDE stream (where the slot is refilled) behind that collective
                if producer is not None:
                    local_returns = self._produce(producer, slot)     # allocated and produced on the side stream
                if local_returns is not slot["stage"]:
                    slot["stage"].copy_(local_returns)
                if producer is None:
                    # the producer's tensor is read on the side stream: keep the caching allocator from recycling it early
                    local_returns.record_stream(self.side)
                self.read_done = self.side.record_event()
                if self.collective:
                    slot["work"] = dist.all_gather_into_tensor(slot["out"], slot["stage"], async_op=True)
                else:
                    slot["out"].copy_(slot["stage"])
        else:
            if producer is not None:
                local_returns = self._produce(producer, slot)
            if local_returns is not slot["stage"]:
                slot["stage"].copy_(local_returns)
            if self.collective:
                parts = [torch.empty_like(slot["stage"]) for _ in range(self.world)]
                dist.all_gather(parts, slot["stage"])
                slot["out"].copy_(torch.cat(parts))
            else:
                slot["out"].copy_(slot["stage"])

    def _produce(self, producer, slot):
        if self._fills:
            producer(slot["stage"])
            return slot["stage"]
        return self._check(producer())

    def close(self):
        if self.rccl is not None:
            if self.side is not None:
                self.side.synchronize()
            self.rccl.close()
            self.rccl = None

    def order_after_read(self, stream=None):
        """Make `stream` (default: the current one) wait until the last launch's source vector has been read."""
        if self.read_done is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(self.read_done)

    def result(self):
        """Block until every launched gather is complete; returns the [world * n_local] tensor of the last one."""
        for slot in self.slots:
            if slot["work"] is not None:
                slot["work"].wait()
                slot["work"] = None
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return (self._last or self.slots[0])["out"]
