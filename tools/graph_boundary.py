"""What does a graph-replay boundary cost on the GPU timeline?  A chain of dependent tiny kernels as (a) one graph of n nodes,
(b) n graphs of one node, (c) eager launches; device time per kernel from events around many repetitions."""
import time

import torch

x = torch.zeros(1024, device="cuda")
s = torch.cuda.Stream()


def bench(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, host * 1e6 / reps


with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1)
    torch.cuda.synchronize()
    for n in (1, 8, 64):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1)
        g.replay()
        d, h = bench(g.replay, 200)
        print("graph of %2d nodes: %.1f us device per replay (%.2f per node), host %.1f us per replay" % (n, d, d / n, h))
    d, h = bench(lambda: x.add_(1), 2000)
    print("eager: %.2f us device per kernel, host %.2f us" % (d, h))
    # a long-running kernel in front, so that the host is certainly ahead: boundary cost seen by the GPU only
    big = torch.zeros(64 * 1024 * 1024, device="cuda")
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s):
        x.add_(1)
    def seq():
        big.add_(1)          # ~100 us
        for _ in range(10):
            g1.replay()
    d, h = bench(seq, 50)
    d0, _ = bench(lambda: big.add_(1), 50)
    print("10 one-node graph replays behind a %.0f us kernel: %.1f us device each (host ahead), host %.1f us per sequence" % (d0, (d - d0) / 10, h))
