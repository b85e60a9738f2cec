"""cfg4: the caption head (30 522 words, 6 layers) on 8 scenes x 8 sentences x 32 tokens, 256 proposals: forward + loss +
backward, eager and replayed from a HIP graph, bf16-operand and exact-fp32 configurations; the fused generator's kernels alone.
    python tools/time_caption.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cap = importlib.import_module("3dvlp_amd.caption")
ml = importlib.import_module("3dvlp_amd.mfma_linear")
ext = importlib.import_module("3dvlp_amd._lib")
an = importlib.import_module("3dvlp_amd.add_norm")
from tests.test_caption import make_endpoints  # noqa: E402
import tests.test_caption as tc  # noqa: E402

tc.V = 30522
dev = torch.device("cuda:0")
torch.manual_seed(0)
head = cap.TransformerDecoderModel(30522).to(dev).train()
e = make_endpoints(8, 8, 256, 32, device="cuda")
e["aggregated_vote_features"].requires_grad_(True)


def step():
    for p in head.parameters():
        p.grad = None
    d = head(dict(e))
    loss, _ = cap.compute_cap_loss(d)
    with ext.deferred_slab_reduce():
        loss.backward()
    return loss


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / reps


for bf in (True, False):
    with ml.bf16_mma(bf):
        for _ in range(3):
            step()
        eager = timed(step, 10)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            step()
        rep = timed(g.replay, 20)
        # the generator alone
        x = torch.randn(1984, 128, device=dev, requires_grad=True)
        tgt = torch.randint(1, 30522, (1984,), device=dev)
        coef = torch.rand(1984, device=dev)

        def gen():
            nll, _ = cap.vocab_nll(x, head.gen_w, head.gen_b, tgt)
            torch.autograd.grad((nll * coef).sum(), (x, head.gen_w, head.gen_b))

        def gen_fwd():
            with torch.no_grad():
                cap.vocab_nll(x, head.gen_w, head.gen_b, tgt)

        tg, tf = timed(gen, 10), timed(gen_fwd, 10)

        def lib_gen():  # what round 2 did: library GEMM + log_softmax + cross entropy on the materialised (1984, 30522) logits
            lp = torch.log_softmax(torch.nn.functional.linear(x, head.gen_w, head.gen_b), -1)
            l_ = torch.nn.functional.nll_loss(lp, tgt, reduction="none")
            torch.autograd.grad((l_ * coef).sum(), (x, head.gen_w, head.gen_b))

        tl = timed(lib_gen, 10)
    flops = 2.0 * 1984 * 128 * 30522
    print(f"{'bf16 operands' if bf else 'exact fp32  '}: caption head fwd+loss+bwd eager {eager:.2f} ms, graph replay {rep:.2f} ms | "
          f"fused generator fwd {tf:.3f} ms ({flops / tf / 1e9:.1f} TFLOP/s), fwd+bwd {tg:.3f} ms ({5 * flops / tg / 1e9:.1f} TFLOP/s "
          f"incl. recomputation) | materialised logits (library GEMM + log_softmax + nll, fwd+bwd) {tl:.3f} ms", flush=True)
