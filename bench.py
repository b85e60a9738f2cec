"""bench.py — scenes/sec of the 3DVLP grounding step (fwd + bwd + all-reduce + AdamW) on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: ScanRefer grounding, 40 000 points, 256 proposals, batch 8 per
GPU (weak scaling: scenes shard across ranks, one flat gradient all-reduce per step), synthetic scenes
(3dvlp_amd/synth.py) resident in HBM before the timed region, random-init weights.
One JSON line on rank 0.  `roofline` is measured live with events on the launch stream around the
hand-written kernels (right after the timed steps, same process and inputs); `cpu_baseline` times the CPU oracle (forward only) on one scene.
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


def time_kernel(fn, reps, inner=8):
    """Mean duration (ms) of `fn` (one hand-written kernel launch): event pairs on the launch stream around `inner`
    back-to-back launches (the kernels serialise on the stream; a pair around a single 30 us kernel would mostly
    measure the launch path), averaged over `reps` such groups."""
    fn()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        e.synchronize()
        total += s.elapsed_time(e)
    return total / (reps * inner)


PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
PMC_KEYS = ("fps_pruned_kernel", "ball_query_kernel<8>", "row_gemm_%s64, 0, 0>", "sdpa_fwd_%s")


def attach_pmc_traffic(kernels, bf):
    """`traffic` = HBM bytes per launch from rocprofv3 PMC passes of the same kernels on the same inputs
    (tools/roofline_kernels.py; FETCH_SIZE and WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot collect counters itself, so it attaches the
    committed measurement (profiles/r01_pmc_*.csv); null when that file is absent."""
    if not os.path.exists(PMC_TRAFFIC):
        return
    table = json.load(open(PMC_TRAFFIC))
    for entry, key in zip(kernels, PMC_KEYS):
        if "row_gemm" in key:
            key = key % ("lds_kernel<" if bf else "kernel<float, ")
        elif "sdpa" in key:
            key = key % ("lds_kernel" if bf else "kernel<false>")
        for name, v in table.items():
            if key in name:
                entry["traffic"] = v["hbm_bytes_corrected"]
                entry["traffic_note"] = "2*FETCH_SIZE + WRITE_SIZE bytes/launch, profiles/r01_pmc_traffic.json"
                break


def report(args, world, elapsed, loss, batch, ext):
    """The JSON line.  Per-kernel roofline entries are measured right after the timed steps, in the same process
    and on the same resident inputs (the steps themselves are hipGraph replays, which cannot be bracketed)."""
    pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    B, n, m = B_PER_GPU, NUM_POINTS, 2048
    bf = args.dtype == "bf16"
    reps = max(3, args.steps)
    pc = batch["point_clouds"]
    xyz = pc[..., :3].contiguous()
    feat_pm = pc[..., 3:].contiguous()

    fps_ms = time_kernel(lambda: pu.furthest_point_sample(xyz, m), reps, inner=1)
    inds = pu.furthest_point_sample(xyz, m)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    bq_ms = time_kernel(lambda: pu.ball_query(0.2, 64, xyz, new_xyz), reps)
    idx = pu.ball_query(0.2, 64, xyz, new_xyz)

    # SA1 layer 1: gather + GEMM (135 -> 64) + BN statistics, the largest grouped-MLP product
    dt = torch.bfloat16 if bf else torch.float32
    C, cout, R = feat_pm.shape[2], 64, B * m * 64
    K1 = (C + 4 + (15 if bf else 7)) // (16 if bf else 8) * (16 if bf else 8)
    W = (torch.randn(cout, K1, device=xyz.device) * 0.05).to(dt)
    Y = torch.empty((R, cout), dtype=dt, device=xyz.device)
    stats = torch.empty((int(ext.load().vlp3d_sa_stat_slabs(R)), 2, cout), dtype=torch.float64, device=xyz.device)
    g_ms = time_kernel(lambda: ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat_pm, B, n, m, 64, C, 0.2, W, K1,
                                        cout, Y, stats, int(bf)), reps)

    # match-module self-attention core: (B*L = 64, 256 queries, 256 keys, 4 heads x 32)
    q = torch.randn(B * LANG_NUM, 256, 128, device=xyz.device)
    att_ms = time_kernel(lambda: fa.sdpa(q, q, q, 4, bf16_mma=bf), reps)

    esz = 2 if bf else 4
    fps_flops = B * (m - 1) * n * 11.0  # 11 flop per distance-update-compare
    fps_peak = PEAK_FP32_VECTOR_TFLOPS * B / NUM_CUS  # one workgroup (CU) per scene
    bq_bytes = B * (12 * n + 12 * m + 4 * m * 64)
    g_bytes = B * n * C * 4 + B * m * 64 * 4 + R * cout * esz + B * n * 12  # features once + idx + Y + xyz
    g_flops = 2.0 * R * (C + 3) * cout
    att_bytes = 4 * q.numel() * 4

    def entry(kernel, bound, work, peak, unit, ms, **extra):
        ach = work / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
        d = {"kernel": kernel, "bound": bound, "achieved": round(ach, 4), "peak": round(peak, 4), "unit": unit,
             "frac": round(ach / peak, 4), "traffic": None, "ms": round(ms, 4)}
        d.update(extra)
        return d

    kernels = [
        entry("fps_pruned_kernel SA1 40000->2048 (bit-exact bounding-box pruned FPS; work = the dense algorithm's "
              "B*(m-1)*n distance-update-compares)", "valu", fps_flops, fps_peak, "TFLOP/s", fps_ms, cus_used=B,
              hbm_algorithmic_GBs=round(B * (12 * n + 4 * m) / (fps_ms * 1e-3) / 1e9, 3),
              streaming_equiv_GBs=round(B * m * n * 20.0 / (fps_ms * 1e-3) / 1e9, 1)),
        entry("ball_query_kernel<8> SA1 r=0.2 ns=64", "hbm", bq_bytes, PEAK_HBM_GBS, "GB/s", bq_ms,
              tests_per_s=round(B * m * n / (bq_ms * 1e-3) / 1e12, 3)),
        entry(("row_gemm_lds_kernel<64,GATHER,STORE>" if bf else "row_gemm_kernel<float,64,GATHER,STORE>") +
              " SA1 layer 1 (gather + 135->64 GEMM + BN sums)", "hbm", g_bytes, PEAK_HBM_GBS,
              "GB/s", g_ms, mfma_TFLOPs=round(g_flops / (g_ms * 1e-3) / 1e12, 2)),
        entry(("sdpa_fwd_lds_kernel (bf16 MFMA, K/V shared through LDS)" if bf else "sdpa_fwd_kernel<fp32 MFMA>") +
              " match self-attention 64x(256x256) h4 d32",
              "hbm", att_bytes, PEAK_HBM_GBS, "GB/s", att_ms, mfma_TFLOPs=round(4.0 * q.shape[0] * 256 * 256 * 128 / (att_ms * 1e-3) / 1e12, 2)),
    ]
    attach_pmc_traffic(kernels, bf)
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
                   "precision": ("bf16 storage + bf16 MFMA (fp32 accumulate) in the grouped per-ball MLPs, bf16 MFMA "
                                 "operands (fp32 I/O, softmax, accumulate) in the attention cores, fp32 elsewhere"
                                 if bf else "fp32 everywhere (exact-fp32 MFMA)"),
                   "launch": "eager" if args.no_graph else "hipGraph replay (fwd+loss+bwd)",
                   "geometry": "inline" if args.no_pipeline else
                   "backbone FPS/ball-query/three_nn of the next batch on a side stream (executed every step)",
                   "loss": float(loss.detach())},
        # dominant hand-written kernel by time (3.6 ms, one workgroup per scene; off the critical path when pipelined)
        "roofline": kernels[0],
        "roofline_kernels": kernels[1:],
    }
    return out


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
    # one process per GPU.  (Rehearsal on a single-GPU box: VLP3D_DIST_BACKEND=gloo lets several ranks share cuda:0.)
    backend = os.environ.get("VLP3D_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step.run(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step.run(batch)
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        out = report(args, world, elapsed, loss, batch, ext)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch_np)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
