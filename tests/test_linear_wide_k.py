"""nn.Linear with 512 input columns on the library's own kernels (mfma_linear._SLICED_K): the captioner's position-wise feed
forward w_2 = Linear(512, 128) (models/caption_module/transformer_captioner.py:95-104) was the one counted library GEMM of the
cfg4 step.  Forward / input gradient take any K % 32 == 0; the weight gradient runs as two 256-column slices of the batched rows
weight gradient when the step's deferred queue is open."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf", [True, False])
def test_linear_512_columns_forward_backward(bf):
    ext = importlib.import_module("3dvlp_amd._lib")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(7)
    R, K, N = 1984, 512, 128
    x = torch.randn(R, K, device="cuda", requires_grad=True)
    lin = torch.nn.Linear(K, N).cuda()
    g = torch.randn(R, N, device="cuda")
    assert ml.supported(x, lin.weight)
    ml.FALLBACKS.clear()
    with ml.bf16_mma(bf):
        y = ml.linear(x, lin.weight, lin.bias)
        with ext.deferred_slab_reduce():
            gx, gw, gb = torch.autograd.grad(y, [x, lin.weight, lin.bias], g)
    torch.cuda.synchronize()
    if bf:
        assert sum(ml.FALLBACKS.values()) == 0   # everything on the library's own kernels inside the queue
    r = (lambda t: t.detach().bfloat16().double()) if bf else (lambda t: t.detach().double())
    tol = dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y.detach().double(), r(x) @ r(lin.weight).t() + lin.bias.detach().double(), **tol)
    torch.testing.assert_close(gx.double(), r(g) @ r(lin.weight), **tol)
    torch.testing.assert_close(gw.double(), r(g).t() @ r(x), rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(gb.double(), r(g).sum(0), rtol=1e-4, atol=1e-3)   # (bf16 configuration: the column sums of the staged, rounded dY)
