"""Are hipMemsetAsync / hipMemcpyAsync nodes of a captured graph ordered against neighbouring kernel nodes and against
the previous replay?  (ROCm 7.2 + torch 2.10 on MI355X.)  Pure torch + one ctypes call into libamdhip64."""
import ctypes, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]


def memset(t):
    rc = hip.hipMemsetAsync(t.data_ptr(), 0, t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def memcpy(dst, src):
    rc = hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), 3,
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for n in (2, 1024, 1 << 18, 1 << 20, 1 << 22):
    x = torch.ones(n, device="cuda"); y = torch.empty(n, device="cuda"); z = torch.empty(n, device="cuda")
    # A: kernel -> memset -> kernel
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.fill_(1.0); memset(x); y.copy_(x * 1.0)
    bad = []
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); bad.append(int((y != 0).sum()))
    print(n, "A kernel->memset->kernel   nonzero:", bad, flush=True)
    # B: memset is the root node; the previous replay ended with a kernel that filled the target
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        memset(x); y.copy_(x * 1.0); x.fill_(1.0)
    bad = []
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); bad.append(int((y != 0).sum()))
    print(n, "B memset(root)->kernel->fill nonzero:", bad, flush=True)
    # B2: the same without a host synchronisation between the replays
    bad = []
    for i in range(4):
        g.replay()
    torch.cuda.synchronize(); bad.append(int((y != 0).sum()))
    print(n, "B2 back-to-back replays     nonzero:", bad, flush=True)
    # C: kernel -> memcpy -> kernel
    x.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.add_(1.0); memcpy(z, x); y.copy_(z * 1.0)
    bad = []
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); bad.append(int((y != float(i + 1)).sum()))
    print(n, "C kernel->memcpy->kernel    wrong:", bad, flush=True)
    # D: memcpy is the root node
    x.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        memcpy(z, x); y.copy_(z * 1.0); x.add_(1.0)
    bad = []
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); bad.append(int((y != float(i)).sum()))
    print(n, "D memcpy(root)->kernel->add wrong:", bad, flush=True)
