"""
Diagnostic (round 3): does a hipMemsetAsync issued while a stream is being captured take effect when the hipGraph is
replayed?  Four cases: issued from the capturing (main) thread or from autograd's worker thread (a custom Function's
backward), 8 or 24 bytes.  Each captured step is  memset(buf, 0) ; buf += 1  on a buffer from the graph's pool: after k
replays buf must read 1 (memset effective) -- k + 1 means the memset node was dropped.
"""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int


def memset0(t, nbytes):
    rc = hip.hipMemsetAsync(C.c_void_p(t.data_ptr()), 0, nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


class Bwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep, nbytes):
        ctx.keep, ctx.nbytes = keep, nbytes
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        buf = torch.empty(ctx.nbytes // 4, device=g.device)
        memset0(buf, ctx.nbytes)
        buf += 1.0
        ctx.keep.append(buf)
        return g * buf.sum() / buf.numel(), None, None


def case(where, nbytes):
    dev = torch.device("cuda", 0)
    keep = []
    x = torch.ones(4, device=dev, requires_grad=True)
    x.grad = torch.zeros(4, device=dev)

    def step():
        if where == "main":
            buf = torch.empty(nbytes // 4, device=dev)
            memset0(buf, nbytes)
            buf += 1.0
            keep.append(buf)
        else:
            Bwd.apply(x, keep, nbytes).sum().backward()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    del keep[:]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    buf = keep[-1]
    buf.fill_(1000.0)                      # what a recycled block of the pool may hold
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append(float(buf[0]))
    print(f"memset from {where:8s} thread, {nbytes:3d} B: buffer after replays {vals} -> "
          f"{'memset effective' if vals == [1.0, 1.0, 1.0] else 'MEMSET NODE NOT EFFECTIVE'}", flush=True)


for where in ("main", "autograd"):
    for nbytes in (8, 24, 4096):
        case(where, nbytes)
