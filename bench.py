"""bench.py — scenes/sec of the 3DVLP grounding step (fwd + bwd + all-reduce + AdamW) on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: ScanRefer grounding, 40 000 points, 256 proposals, batch 8 per
GPU (weak scaling: scenes shard across ranks, one flat gradient all-reduce per step), synthetic scenes
(3dvlp_amd/synth.py) resident in HBM before the timed region, random-init weights.
One JSON line on rank 0.  `roofline` is measured live with events on the launch stream around the
dominant hand-written kernel; `cpu_baseline` times the CPU oracle (forward only) on one scene.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 8
NUM_POINTS = 40000
LANG_NUM = 8
# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PEAK_FP32_VECTOR_TFLOPS = 157.3
NUM_CUS = 256


class KernelTimer:
    """HIP-event pairs on torch's current stream (the stream the C ABI launches on) around one op."""

    def __init__(self, module, fn_name, pick):
        self.module, self.fn_name, self.pick = module, fn_name, pick
        self.orig = getattr(module, fn_name)
        self.events = []
        self.enabled = False
        setattr(module, fn_name, self._wrapped)

    def _wrapped(self, *a, **k):
        if not (self.enabled and self.pick(*a, **k)):
            return self.orig(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = self.orig(*a, **k)
        e.record()
        self.events.append((s, e))
        return out

    def mean_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.events) / max(1, len(self.events))


def cpu_baseline(batch_np):
    from oracle import baseline
    import numpy as np
    pc = batch_np["point_clouds"][:1]
    xyz = np.ascontiguousarray(pc[..., :3])
    feats = np.ascontiguousarray(pc[..., 3:].transpose(0, 2, 1))
    sec, parts = baseline.scene_forward(xyz, feats, LANG_NUM)
    return {"value": round(1.0 / sec, 4), "unit": "scenes/s", "cores": baseline.threads_used(), "kind": "port",
            "sample": "1 scene (40k pts, 256 proposals, 8 sentences), FORWARD ONLY (the oracle has no backward): "
                      "C/OpenMP geometry + numpy dense; seconds per part: " +
                      ", ".join(f"{k} {v:.2f}" for k, v in parts.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="compute the backbone geometry (FPS / ball query) inline instead of one batch ahead")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    synth = importlib.import_module("3dvlp_amd.synth")
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    ddp = importlib.import_module("3dvlp_amd.ddp")
    ext = importlib.import_module("3dvlp_amd._lib")

    first, _ = ddp.shard_range(B_PER_GPU * world, rank, world)
    batch_np = synth.make_batch(first, B_PER_GPU, NUM_POINTS, LANG_NUM)
    batch = gs.batch_to_device(batch_np, device)
    step = gs.GroundingStep(device, epoch=50, sa_dtype=torch.bfloat16 if args.dtype == "bf16" else None,
                            use_graph=not args.no_graph, pipeline=not args.no_pipeline)
    ddp.broadcast_parameters(step.model)

    # dominant hand-written kernel: FPS of SA1 (40 000 -> 2048)
    fps_timer = KernelTimer(ext, "furthest_point_sampling", lambda pts, m: pts.shape[1] == NUM_POINTS)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step.run(batch)
    sync()
    fps_timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step.run(batch)
    sync()
    elapsed = time.perf_counter() - t0
    fps_timer.enabled = False
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        if not fps_timer.events:
            # under hipGraph replay the Python launch wrapper is not executed, so the kernel cannot be bracketed
            # inside the timed steps: launch the same kernel on the same inputs and stream right after them
            pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
            xyz = batch["point_clouds"][..., :3].contiguous()
            fps_timer.enabled = True
            for _ in range(max(3, args.steps)):
                pu.furthest_point_sample(xyz, 2048)
            torch.cuda.synchronize()
        fps_ms = fps_timer.mean_ms()
        m, n = 2048, NUM_POINTS
        flops = B_PER_GPU * (m - 1) * n * 11.0  # 11 flop per distance-update-compare
        peak = PEAK_FP32_VECTOR_TFLOPS * B_PER_GPU / NUM_CUS  # one workgroup (CU) per scene
        achieved = flops / (fps_ms * 1e-3) / 1e12
        out = {
            "metric": "scenes/sec fwd+bwd, 40k-pt/256-proposal grounding",
            "value": round(B_PER_GPU * world * args.steps / elapsed, 3),
            "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg2: ScanRefer grounding step, 40k pts, 256 proposals, 8 sentences/scene",
                       "batch_per_gpu": B_PER_GPU, "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}",
                       "step": "fwd + reduced loss + bwd + flat grad all-reduce + AdamW",
                       "precision": ("bf16 storage + bf16 MFMA (fp32 accumulate) in the grouped per-ball MLPs, fp32 elsewhere"
                                     if args.dtype == "bf16" else "fp32 everywhere (exact-fp32 MFMA)"),
                       "launch": "eager" if args.no_graph else "hipGraph replay (fwd+loss+bwd)",
                       "geometry": "inline" if args.no_pipeline else
                       "backbone FPS/ball-query/three_nn of the next batch on a side stream (executed every step)", "loss": float(loss.detach())},
            "roofline": {"kernel": "fps_kernel<1024,24,9> SA1 40000->2048", "bound": "valu",
                         "achieved": round(achieved, 4), "peak": round(peak, 4), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None, "ms": round(fps_ms, 4),
                         "cus_used": B_PER_GPU,
                         "hbm_algorithmic_GBs": round(B_PER_GPU * (12 * n + 4 * m) / (fps_ms * 1e-3) / 1e9, 3),
                         "streaming_equiv_GBs": round(B_PER_GPU * m * n * 20.0 / (fps_ms * 1e-3) / 1e9, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch_np)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
