"""hipMemsetAsync inside a captured hipGraph on this build (ROCm 7.2, PyTorch 2.10): does the memset node do what the eager call does?
(round 6; profiles/r06_td3_hipgraph_learning.txt).  torch's reductions initialise their cross-block semaphores with cudaMemsetAsync
(ATen/native/cuda/Reduce.cuh), so a broken memset node makes captured multi-block reductions -- every bias gradient -- unreliable."""
import ctypes as C
import torch
dev = "cuda:0"
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
hip = C.CDLL(path)
hip.hipMemsetAsync.restype = C.c_int
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
def trial(nbytes, value, graph):
    n = nbytes // 4 + 4
    buf = torch.zeros(n, dtype=torch.int32, device=dev)
    out = torch.zeros(n, dtype=torch.int32, device=dev)
    def body():
        buf.fill_(0x07070707)
        rc = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), value, nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        out.copy_(buf)
    want = torch.full((n,), 0x07070707, dtype=torch.int32)
    v4 = value | (value << 8) | (value << 16) | (value << 24)
    if v4 >= 1 << 31:
        v4 -= 1 << 32
    want[:nbytes // 4] = v4
    bad = 0
    if graph:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
    for it in range(50):
        out.fill_(-1)
        g.replay() if graph else body()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out.cpu(), want))
    return bad, out[:6].cpu().tolist(), want[:6].tolist()
for nbytes in (4, 64, 1024, 4096, 65536):
    for value in (0, 0x5A):
        e = trial(nbytes, value, False)
        g = trial(nbytes, value, True)
        print("memset %6d bytes of 0x%02x: eager wrong %2d / 50 | captured wrong %2d / 50   captured result %s expected %s" % (nbytes, value, e[0], g[0], g[1], g[2]), flush=True)
