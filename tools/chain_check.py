"""Row chain (csrc/rows_chain.hip) against the unfused modules: forward / gradient differences and launch times."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib

pkg = importlib.import_module("3dvlp_amd")
from importlib import import_module

add_norm = import_module("3dvlp_amd.add_norm")
mfma_linear = import_module("3dvlp_amd.mfma_linear")
row_chain = import_module("3dvlp_amd.row_chain")

torch.manual_seed(0)
dev = "cuda"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p = 0.1
lin = lambda n, k: torch.nn.Linear(k, n).to(dev)
fo, l1, l2, nx = lin(128, 128), lin(256, 128), lin(128, 256), lin(384, 128)
m1, m2 = lin(128, 128), lin(128, 128)
n1, n2 = torch.nn.LayerNorm(128).to(dev), torch.nn.LayerNorm(128).to(dev)
for n in (n1, n2):
    n.weight.data.uniform_(0.5, 1.5)
    n.bias.data.uniform_(-0.5, 0.5)
a0 = torch.randn(R, 128, device=dev)
x0 = torch.randn(R, 128, device=dev)
g3 = torch.randn(R, 128, device=dev)
gq = torch.randn(R, 384, device=dev)
gm = torch.randn(R, 128, device=dev)
params = [q for m in (fo, l1, l2, nx, m1, m2, n1, n2) for q in m.parameters()]


def unfused(a, x):
    y = mfma_linear.linear(a, fo.weight, fo.bias)
    x1 = add_norm.add_norm(x, y, n1, p, True)
    z = mfma_linear.linear(x1, l1.weight, l1.bias)
    h = add_norm.act_dropout(z, "relu", p, True)
    f = mfma_linear.linear(h, l2.weight, l2.bias)
    x3 = add_norm.add_norm(x1, f, n2, p, True)
    qkv = mfma_linear.linear(x3, nx.weight, nx.bias)
    m = add_norm.act_dropout(mfma_linear.linear(x3, m1.weight, m1.bias), "gelu", 0.5, True)
    return x1, x3, qkv, m


def chained(a, x):
    t = row_chain.run(a, [row_chain.linear_add_norm(fo.weight, fo.bias, n1, x, p), row_chain.linear(l1.weight, l1.bias, "relu", p),
                          row_chain.linear_add_norm(l2.weight, l2.bias, n2, ("tile", 1), p), row_chain.linear(nx.weight, nx.bias)])
    m = add_norm.act_dropout(mfma_linear.linear(t[2], m1.weight, m1.bias), "gelu", 0.5, True)
    return t[0], t[2], t[3], m


def go(fn):
    add_norm._CALLS[0] = 100
    a, x = a0.clone().requires_grad_(), x0.clone().requires_grad_()
    for q in params:
        q.grad = None
    with mfma_linear.bf16_mma(True):
        x1, x3, qkv, m = fn(a, x)
        ((x3 * g3).sum() + (qkv * gq).sum() + (m * gm).sum()).backward()
    return [x1, x3, qkv, m], [a.grad, x.grad] + [q.grad for q in params]


fu, gu = go(unfused)
fc, gc = go(chained)
for name, u, c in zip(("x1", "x3", "qkv", "m"), fu, fc):
    d = (u - c).abs()
    print("fwd %-4s max %.3e mean %.3e (scale %.2f)" % (name, d.max().item(), d.mean().item(), u.abs().mean().item()))
for i, (u, c) in enumerate(zip(gu, gc)):
    if u is None:
        continue
    d = (u - c).abs()
    print("grad %2d max %.3e mean %.3e (scale %.3f)" % (i, d.max().item(), d.mean().item(), u.abs().mean().item()))


def timeit(fn, n=30):
    with torch.no_grad(), mfma_linear.bf16_mma(True):
        for _ in range(5):
            fn(a0, x0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn(a0, x0)
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3


print("unfused %.1f us   chained %.1f us  (forward only, graph replay)" % (timeit(unfused), timeit(chained)))


def chain_only(k):
    def f(a, x):
        st = [row_chain.linear_add_norm(fo.weight, fo.bias, n1, x, p), row_chain.linear(l1.weight, l1.bias, "relu", p),
              row_chain.linear_add_norm(l2.weight, l2.bias, n2, ("tile", 1), p), row_chain.linear(nx.weight, nx.bias)][:k]
        return row_chain.run(a, st)
    return f


for k in (1, 2, 3, 4):
    print("chain stages 0..%d: %.1f us" % (k - 1, timeit(chain_only(k))))
print("plain linear 128->384: chain %.1f us, linear_fwd %.1f us" % (
    timeit(lambda a, x: row_chain.run(a, [row_chain.linear(nx.weight, nx.bias)])),
    timeit(lambda a, x: mfma_linear.linear(a, nx.weight, nx.bias))))
print("plain linear 128->128: chain %.1f us, linear_fwd %.1f us" % (
    timeit(lambda a, x: row_chain.run(a, [row_chain.linear(m1.weight, m1.bias)])),
    timeit(lambda a, x: mfma_linear.linear(a, m1.weight, m1.bias))))
