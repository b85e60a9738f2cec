"""bench.py prints ONE JSON line with the fields the driver parses (run on the GPU box as a subprocess)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "scenes/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "reference's loss" in d["config"]["step"]
    sm = d["step_ms"]
    # two timed steps right after the capture: the wall-clock mean may contain a slow first replay, the per-step events
    # may not exceed it by much
    assert 0 < sm["p10"] <= sm["median"] <= sm["p90"] and sm["median"] < 1.5 * d["ms_per_step"]
    rs = d["roofline_step"]
    assert rs["flops_per_step"] > 3e11 and 0 < rs["frac_mfma"] < 1 and 0 < rs["frac_hbm"] < 1
    # the headline is the kernel that dominates the rocprofv3 summary: SA1's pruned FPS, against the VALU rate of its 8 CUs
    assert "fps_pruned" in r["kernel"] and r["bound"] == "valu" and r["cus_used"] == 8 and r["ms_is"].startswith("in-step")
    names = " ".join(k["kernel"] for k in d["roofline_kernels"])
    assert "MAIN-stream" in names and "ball query" in names and "cross-attention" in names
    for key in ("ms_per_step_padded", "ms_per_step_fp32", "ms_per_step_host_batches"):
        assert d[key] > 0.5 * d["step_ms"]["median"], key
    assert d["allreduce_exposed_ms"] is None      # one GPU: no collective
    hw = d["hw"]
    assert 2000 < hw["hbm_read_GBs"] < 9000 and 500 < hw["bf16_mfma_TFLOPs"] < 3000 and 50 < hw["fp32_fma_TFLOPs"] < 200


@pytest.mark.gpu
def test_bench_two_ranks_gloo_on_one_gpu():
    """The data-parallel path end to end on the hardware we do have: two fresh ranks (torch.distributed.run children,
    started before they touch the GPU) share cuda:0, gradients go through the flat-bucket all-reduce (gloo stands in for
    RCCL on a one-GPU box), and after the steps both replicas hold identical parameters."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VLP3D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-kernels", "--check-replicas"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert d["allreduce_exposed_ms"] is not None and d["allreduce_exposed_ms"] >= 0 and "two pieces" in d["allreduce"]
    a, b = d["replica_param_checksums"]
    assert a == b, (a, b)                      # identical replicas after averaged-gradient steps
    assert d["config"]["loss"] == d["config"]["loss"] and d["config"]["loss"] > 0   # finite
