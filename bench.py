"""bench.py — scenes/sec of the 3DVLP grounding step (fwd + reference loss + bwd + all-reduce + AdamW) on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]            (N=1, K=100, W=20)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: ScanRefer grounding, 40 000 points, 256 proposals, batch 8 per GPU (weak
scaling: scenes shard across ranks, one flat gradient all-reduce per step), synthetic scenes (3dvlp_amd/synth.py)
resident in HBM before the timed region, random-init weights.  ONE JSON line on rank 0:

  value / ms_per_step   K steps between two barrier + synchronize brackets (max over ranks) — the contract's number
  step_ms               per-step durations from events on the launch stream: median, p10, p90
  roofline              the kernel that dominates the rocprofv3 summary of this command: SA1's pruned FPS (side stream, ~30 % of all
                        kernel time), against the fp32 VALU rate of the 8 CUs it occupies (`ms`: device-clock stamps captured
                        around the entry point inside an instrumented copy of the replayed step, csrc/hwprobe.hip vlp3d_stamp;
                        `ms_isolated`: the same launch replayed alone); `traffic` = PMC HBM bytes of the same launch
                        (profiles/r0X_pmc_traffic.json, produced by tools/pmc_traffic.py; null when absent)
  roofline_kernels      the main-stream candidates (SA1 gather GEMM / its weight gradient / layer-3 input gradient; the largest by
                        in-step duration is marked), relation-bias backward, grid ball query, row chains, attention cores (self +
                        cross; algorithmic_bytes = SURVEY.md §8(d)'s bf16 figure), each with `ms` in-step and `ms_isolated`
  ms_per_step_padded / ms_per_step_fp32 / ms_per_step_host_batches
                        the same step without the distinct-row evaluation / in the 1e-4 parity configuration / fed from pinned
                        host memory (PCIe-inclusive) — never `value`
  padded_form_ms_per_step / linear_library_fallbacks_per_step / stamp_gap_us / empty_kernel_in_step_us
                        what the headline does not show (the last: an empty kernel timed the same way = the launch floor)
  roofline_step         whole step: algorithmic flops and bytes per step / ms_per_step against the chip's peaks
  hw                    this box's measured denominators (HBM read, bf16 MFMA, fp32 FMA) beside the guide's figures
  cpu_baseline          the same training step (fwd + loss + bwd + AdamW) on the host CPU (oracle/baseline.py)
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
NUM_TOKENS = 49
# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_FP32_VECTOR_TFLOPS = 157.3
NUM_CUS = 256
PMC_TRAFFIC = next((p for p in (os.path.join(ROOT, "profiles", "r04_pmc_traffic.json"),
                                os.path.join(ROOT, "profiles", "r03_pmc_traffic.json"),
                                os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r04_pmc_traffic.json"))


def time_kernel(fn, reps, inner=8):
    """Mean duration (ms) of `fn` (one hand-written kernel launch, or the few launches of one entry point): `inner`
    back-to-back launches are captured into a HIP graph — the way the step itself issues them — and event pairs on the
    launch stream bracket each replay; averaged over `reps` replays.  (Issued from Python one by one, a 20 us kernel is
    host-bound: the pair would mostly measure the launch path.)"""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        e.synchronize()
        total += s.elapsed_time(e)
    return total / (reps * inner)


def measure_hw(ext, device):
    """the denominators re-measured on this box (csrc/hwprobe.hip)."""
    sink = torch.zeros(1, device=device)
    buf = torch.empty(1 << 30, dtype=torch.uint8, device=device)  # 1 GiB
    buf.zero_()
    dst = torch.empty_like(buf)
    blocks = NUM_CUS * 8
    rd = time_kernel(lambda: ext.call("vlp3d_probe_read", buf, buf.numel(), blocks, sink), 3, inner=4)
    cp = time_kernel(lambda: dst.copy_(buf), 3, inner=4)
    it = 2048
    mf = time_kernel(lambda: ext.call("vlp3d_probe_mfma_bf16", it, blocks, sink), 3, inner=2)
    fm = time_kernel(lambda: ext.call("vlp3d_probe_fma_f32", 4 * it, blocks, sink), 3, inner=2)
    return {"hbm_read_GBs": round(buf.numel() / (rd * 1e-3) / 1e9, 1),
            "hbm_copy_GBs_read_plus_write": round(2 * buf.numel() / (cp * 1e-3) / 1e9, 1),
            "bf16_mfma_TFLOPs": round(blocks * 4 * it * 4 * 2 * 32 * 32 * 16 / (mf * 1e-3) / 1e12, 1),
            "fp32_fma_TFLOPs": round(blocks * 256 * 4 * it * 16 / (fm * 1e-3) / 1e12, 1),
            "guide": {"hbm_GBs": PEAK_HBM_GBS, "bf16_mfma_TFLOPs": PEAK_BF16_MFMA_TFLOPS,
                      "fp32_vector_TFLOPs": PEAK_FP32_VECTOR_TFLOPS},
            "note": "1 GiB streaming read / device copy; 4 independent v_mfma_f32_32x32x16_bf16 chains per wave, 8 waves "
                    "per SIMD-group; 8 independent v_fma_f32 chains per lane.  Fractions below use the guide's peaks."}


def step_work(B, esz):
    """Algorithmic work of ONE step on B scenes: dense flops (forward; fwd+bwd = 3x) and the HBM bytes
    a fused implementation with stored pre-activations has to move (every stored tensor written once and read once
    forward, read once more and its gradient written + read once backward)."""
    L, K, T = LANG_NUM, 256, NUM_TOKENS
    sa = [(B * 2048 * 64, [135, 64, 64, 128], 40000, 132), (B * 1024 * 32, [131, 128, 128, 256], 2048, 128),
          (B * 512 * 16, [259, 128, 128, 256], 1024, 256), (B * 256 * 16, [259, 128, 128, 256], 512, 256),
          (B * 256 * 16, [259, 128, 128, 128], 1024, 256)]
    flops = 0.0
    byts = 0.0
    for R, dims, n_in, c_in in sa:
        flops += 2.0 * R * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        y = R * sum(dims[1:]) * esz
        fwd = B * n_in * c_in * 4 + R * 4 + 2 * y          # features once + ball-query idx + Y written, read
        byts += fwd + (2 * y + B * n_in * c_in * 4)        # backward: Y read again, G written+read folded in 2y, dfeat
    rows = lambda r, dims: 2.0 * r * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    flops += rows(B * 512, [512, 256, 256]) + rows(B * 1024, [512, 256, 256])      # FP1, FP2
    flops += rows(B * 1024, [256, 256, 256, 259])                                    # voting
    flops += rows(B * K, [128, 128, 128, 28])                                        # ROI heads
    flops += rows(B * K, [128, 128, 128])                                            # relation features_concat
    for _ in range(2):                                                               # relation layers
        flops += rows(B * K, [128, 128]) + rows(B * K, [27, 128]) + 4 * rows(B * K, [128, 128])
        flops += 2 * 2.0 * B * K * K * 128 + rows(B * K * K, [4, 32, 32, 4])
    for _ in range(2):                                                               # match decoder layers
        r = B * L * K
        flops += 4 * rows(r, [128, 128]) + 2 * 2.0 * B * L * K * K * 128            # self attention
        flops += 2 * rows(r, [128, 128]) + 2 * rows(B * L * T, [128, 128]) + 2 * 2.0 * B * L * K * T * 128
        flops += rows(r, [128, 256, 128])
        byts += 4 * r * 128 * 4 * 2 + (2 * r + 2 * B * L * T) * 128 * 4 * 2        # SDPA q,k,v,o once each way
    flops += rows(B * L * K, [128, 128, 128, 1])                                     # match MLP
    flops += rows(B * L + 2 * B * K, [128, 128])                                     # contrast projections
    return 3.0 * flops, byts


def pmc_traffic():
    return json.load(open(PMC_TRAFFIC)) if os.path.exists(PMC_TRAFFIC) else {}


STAMPED = ("vlp3d_sa_fwd_gather", "vlp3d_sa_fwd_layer", "vlp3d_sa_bwd_layer", "vlp3d_sa_bwd_gather", "vlp3d_sa_wgrad",
           "vlp3d_sa_pool", "vlp3d_sdpa_fwd", "vlp3d_sdpa_bwd", "vlp3d_relation_bias_fwd", "vlp3d_relation_bias_bwd",
           "vlp3d_furthest_point_sampling_pruned", "vlp3d_ball_query_grid", "vlp3d_ball_query_sorted", "vlp3d_sa_last_dgrad",
           "vlp3d_sa_last_wgrad", "vlp3d_rows_chain_io", "vlp3d_rows_chain_bwd", "vlp3d_sdpa_fwd_io", "vlp3d_sdpa_bwd_io",
           "vlp3d_probe_empty")


def in_step_durations(args, batch, gs, ext, steps=12, side_stream=None):
    """Durations of the named entry points INSIDE the replayed step: a second, instrumented copy of the step is captured
    with one-thread clock stamps (csrc/hwprobe.hip vlp3d_stamp) bracketing every such call, replayed `steps` times, and the
    stamp slots of the last replays are read back.  Returns ([(entry, int/float args, median us)], floor_us,
    linear fall-backs per step).  floor_us: the same bracket around an empty kernel (two dispatch gaps), to be subtracted."""
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    dev = batch["point_clouds"].device
    step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16 if args.dtype == "bf16" else None, use_graph=True, pipeline=True,
                            side_stream=side_stream)
    with ext.Stamps(STAMPED, dev, capacity=1024) as st:
        def begin():
            st.log.clear()
            ml.FALLBACKS.clear()
            st.calibrate_pending = True   # one bare stamp pair (a dispatch gap) + a bracketed empty kernel, captured in front of
            #                               the first stamped call
        step.on_capture = begin
        step.run(batch)
        fallbacks = sum(ml.FALLBACKS.values())
        samples = []
        for _ in range(steps):
            step.run(batch)
            torch.cuda.synchronize()
            samples.append([d for _, _, d in st.durations()])
        log = list(st.log)
    med = [sorted(col)[len(col) // 2] for col in zip(*samples[2:])]
    gap = next(m for (n, _), m in zip(log, med) if n == "stamp_gap")
    empty = next(m for (n, _), m in zip(log, med) if n == "vlp3d_probe_empty")
    del step
    return [(n, a, m) for (n, a), m in zip(log, med) if n not in ("vlp3d_probe_empty", "stamp_gap")], (gap, empty), fallbacks


def kernel_rooflines(args, batch, ext, gs, side_stream=None):
    """Per-kernel roofline entries.  `ms` of every entry is the kernel's duration INSIDE the replayed step (in_step_durations:
    clock stamps captured around the call in an instrumented copy of the step; the bracket's own cost, measured around an
    empty kernel, is subtracted and reported as `bracket_us`); `ms_isolated` is the same entry point replayed alone from a
    graph of 8 back-to-back launches with events on the launch stream, inputs warm.  `achieved` / `frac` use `ms`.
    The headline `roofline` is the hand-written main-stream kernel with the largest in-step duration."""
    pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    B, n, m = B_PER_GPU, NUM_POINTS, 2048
    bf = args.dtype == "bf16"
    reps = 5
    pc = batch["point_clouds"]
    xyz = pc[..., :3].contiguous()
    feat_pm = pc[..., 3:].contiguous()
    table = pmc_traffic()
    stamped, (gap_us, empty_us), fallbacks = in_step_durations(args, batch, gs, ext, side_stream=side_stream)
    bracket_us = 2.0 * gap_us   # stamp -> kernel -> stamp = gap + duration + gap

    def in_step(name, pred=lambda a: True, pick=max):
        hits = [us for nme, a, us in stamped if nme == name and pred(a)]
        return (pick(hits) - bracket_us) * 1e-3 if hits else None

    def entry(kernel, key, bound, work, peak, unit, ms_iso, ms_step, **extra):
        ms = ms_step if ms_step is not None else ms_iso
        if ms is None:   # the entry point is not part of this configuration's step (e.g. csrc/sa_last.hip under --dtype fp32)
            return {"kernel": kernel, "ms": None}
        ach = work / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
        d = {"kernel": kernel, "bound": bound, "achieved": round(ach, 4), "peak": round(peak, 4), "unit": unit,
             "frac": round(ach / peak, 4), "traffic": None, "ms": round(ms, 4), "ms_isolated": round(ms_iso, 4) if ms_iso else None,
             "ms_is": "in-step (clock stamps inside the replayed graph)" if ms_step is not None else "isolated replay"}
        keys = () if not key else ((key,) if isinstance(key, str) else tuple(key))
        hits = [v["hbm_bytes_corrected"] for name, v in table.items() if any(k in name for k in keys)]
        if hits:  # a tuple of keys = a multi-kernel entry (the grid ball query): its launches' traffic summed
            d["traffic"] = sum(hits) if len(keys) > 1 else hits[0]
            d["traffic_note"] = "2*FETCH_SIZE + WRITE_SIZE bytes/launch, profiles/" + os.path.basename(PMC_TRAFFIC)
        d.update(extra)
        return d

    fps_ms = time_kernel(lambda: pu.furthest_point_sample(xyz, m), reps, inner=1)
    inds = pu.furthest_point_sample(xyz, m)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    _, fps_ws = ext.furthest_point_sampling(xyz, m, "pruned", return_workspace=True)
    bq_ms = time_kernel(lambda: ext.ball_query_sorted(new_xyz, xyz, 0.2, 64, fps_ws), reps)
    bq_grid_ms = time_kernel(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64, "grid"), reps)
    idx = ext.ball_query_sorted(new_xyz, xyz, 0.2, 64, fps_ws)

    # SA1 layer 1: gather + GEMM (135 -> 64) + BN statistics
    dt = torch.bfloat16 if bf else torch.float32
    C, cout, R = feat_pm.shape[2], 64, B * m * 64
    K1 = (C + 4 + (15 if bf else 7)) // (16 if bf else 8) * (16 if bf else 8)
    W = (torch.randn(cout, K1, device=xyz.device) * 0.05).to(dt)
    Y = torch.empty((R, cout), dtype=dt, device=xyz.device)
    stats = torch.empty((int(ext.load().vlp3d_sa_stat_slabs(R)), 2, cout), dtype=torch.float64, device=xyz.device)
    # in the bf16 configuration the step evaluates the grouped MLP on the DISTINCT rows of every ball (csrc/sa_compact.hip:
    # ball-query padding removed); the kernel is measured the way the step runs it
    compact = bf and os.environ.get("VLP3D_SA_COMPACT", "1") != "0"
    cm = (None, None, 0)
    rows = R
    if compact:
        rowptr, crow = ext.sa_compact(idx, n)
        cm = (crow, rowptr, B * m)
        rows = (int(rowptr[-1]) + 31) // 32 * 32
    fbit = 0
    if bf and "k/feat_bf" in batch:   # what the step's SA1 reads: the loader's bf16 rows
        feat_pm, fbit = batch["k/feat_bf"], 2
    g_ms = time_kernel(lambda: ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat_pm, B, n, m, 64, C, 0.2, W, K1,
                                        cout, Y, stats, int(bf) | fbit, *cm), reps)

    # match-module attention cores: (B*L = 64, 256 queries, 4 heads x 32): self 256 keys, cross 49 keys
    BL = B * LANG_NUM
    q = torch.randn(BL, 256, 128, device=xyz.device)
    kc = torch.randn(BL, NUM_TOKENS, 128, device=xyz.device)
    if bf:  # what the step's match decoder launches: the cores on bf16 rows (merged q|k|v; q with the tokens' fp32 k|v)
        qkv16 = torch.randn(BL, 256, 384, device=xyz.device).bfloat16()
        q16, kvc = qkv16[..., :128].contiguous(), torch.randn(BL, NUM_TOKENS, 256, device=xyz.device).bfloat16()
        att_ms = time_kernel(lambda: ext.sdpa_fwd_rows(qkv16[..., :128], qkv16[..., 128:256], qkv16[..., 256:], 4, None, True), reps)
        xat_ms = time_kernel(lambda: ext.sdpa_fwd_rows(q16, kvc[..., :128], kvc[..., 128:], 4, None, True), reps)
    else:
        att_ms = time_kernel(lambda: fa.sdpa(q, q, q, 4, bf16_mma=bf), reps)
        xat_ms = time_kernel(lambda: fa.sdpa(q, kc, kc, 4, bf16_mma=bf), reps)

    esz = 2 if bf else 4
    is_sa1 = lambda a: 40000 in a and 2048 in a
    fps_flops = B * (m - 1) * n * 11.0  # 11 flop per distance-update-compare of the DENSE algorithm
    fps_peak = PEAK_FP32_VECTOR_TFLOPS * B / NUM_CUS  # one workgroup (CU) per scene
    bq_bytes = B * (12 * n + 12 * m + 4 * m * 64)
    # features once + row map (idx, or the 16-byte compact entries) + Y + xyz, for the rows the kernel evaluates
    fsz = 2 if fbit else 4   # the feature rows as the step reads them (bf16 rows from the loader in the bf16 configuration)
    g_bytes = B * n * C * fsz + (rows * 16 if compact else B * m * 64 * 4) + rows * cout * esz + B * n * 12
    g_bytes_dense = B * n * C * fsz + B * m * 64 * 4 + R * cout * esz + B * n * 12
    g_flops = 2.0 * rows * (C + 3) * cout
    # Q, K, V, O touched once in bf16 = the ALGORITHMIC bytes.  bf16 configuration since round 4: the cores read
    # and write bf16 rows — q, the tokens' k|v, the merged q|k|v, out: moved = algorithmic; fp32 configuration: fp32 rows
    att_alg, att_moved = 4 * q.numel() * 2, 4 * q.numel() * (2 if bf else 4)
    xat_alg = (2 * q.numel() + 2 * kc.numel()) * 2
    xat_moved = xat_alg if bf else 2 * xat_alg
    # SA1 layer-1 weight gradient: dW1 = dY1^T A0 over the evaluated rows: features once, G1 and Y1 rows once, the row map
    wg_bytes = B * n * C * fsz + rows * (2 * cout * esz + 16)
    # SA1 layer-3 input gradient (vlp3d_sa_bwd_layer, pooled-gradient loader + mask epilogue): Y3 and Y2 rows once, G2 written,
    # the row map, the pooled gradient / arg-max tensors of the balls
    dg_bytes = rows * (2 * 64 * esz + 16) + B * m * 128 * 5          # since round 4: Y2 in, G2 out, row map, pooled rows (no Y3)
    wg3_bytes = rows * (64 * esz + 16) + B * m * 128 * 5 + 512 * 128 * 64 * 4   # Y2, row map, pooled rows, the dW3 slabs
    rel_pairs = B * 256 * 256
    rel_flops = rel_pairs * 2.0 * (4 * 32 + 32 * 32 + 32 * 4) * 3   # forward recomputation + both backward products per pair
    gname = ("row_gemm_lds_kernel<64,GATHER,STORE>" if bf else "row_gemm_kernel<float,64,GATHER,STORE>")
    sdpa_name = ("sdpa_fwd_lds_kernel (bf16 MFMA, bf16 rows)" if bf else "sdpa_fwd_kernel<fp32 MFMA>")
    sdpa_entry = "vlp3d_sdpa_fwd_io" if bf else "vlp3d_sdpa_fwd"
    cands = [
        entry(gname + " SA1 layer 1 (gather + 135->64 GEMM + BN sums)",
              "row_gemm_lds_kernel<64, 0, 0" if bf else "row_gemm_kernel<float, 64, 0, 0>", "hbm", g_bytes,
              PEAK_HBM_GBS, "GB/s", g_ms, in_step("vlp3d_sa_fwd_gather", is_sa1),
              mfma_TFLOPs=round(g_flops / (g_ms * 1e-3) / 1e12, 2), algorithmic_bytes=g_bytes, rows_evaluated=rows, rows_padded=R,
              padded_form_bytes=g_bytes_dense,
              note=("distinct rows of every ball only (ball-query padding removed, DESIGN.md §4.6): %.1f %% of the padded "
                    "rows; `achieved` counts the bytes of the rows evaluated" % (100.0 * rows / R)) if compact else "padded rows"),
        entry("wgrad_kernel<bf16,...,GATHER> SA1 layer 1 weight gradient (dY1^T x gathered rows)", "::wgrad_kernel<__hip_bfloat16, 64, 0,", "hbm", wg_bytes,
              PEAK_HBM_GBS, "GB/s", None, in_step("vlp3d_sa_wgrad", is_sa1), algorithmic_bytes=wg_bytes),
        entry("sa_last_dgrad_kernel<64,128> SA1 layer 3 input gradient WITHOUT the layer's pre-activation (csrc/sa_last.hip: "
              "(k1 G) W3 - w (W3^T alpha + a2 Q), ReLU mask + BN-backward sums; round 3: row_gemm_lds_kernel<64,BNBWD,MASK> read Y3)",
              "sa_last_dgrad_kernel<64, 128>", "hbm", dg_bytes, PEAK_HBM_GBS, "GB/s", None,
              in_step("vlp3d_sa_last_dgrad", lambda a: a[0] == B * m and a[2] == 64), algorithmic_bytes=dg_bytes,
              note="Y2 rows once, G2 written, the row map, the balls' pooled gradient / arg-max rows; runs beside the side stream's "
                   "deferred weight-gradient graph"),
        entry("sa_last_wgrad_kernel<64,128> SA1 layer 3 weight gradient WITHOUT Y3 ((k1 G)^T a2 - alpha (x) s - diag(beta) W3 M)",
              "sa_last_wgrad_kernel<64, 128>", "hbm", wg3_bytes, PEAK_HBM_GBS, "GB/s", None,
              in_step("vlp3d_sa_last_wgrad", lambda a: a[0] == B * m and a[2] == 64), algorithmic_bytes=wg3_bytes),
    ]
    cands = [c for c in cands if c["ms"] is not None]
    Rm = BL * 256
    # a (bf16 rows from the core), res | x2, xhat2, x3, xhat3 (fp32), the FFN stage's h and q|k|v (bf16 rows; no pre-activation)
    chain_bytes = Rm * (2 * 128 + 4 * (128 + 128 + 128 + 128 + 128) + 2 * 256 + 2 * 384)
    # d x3 (+ base), xhat3, xhat2 (fp32), h (bf16 rows, in place of z) | dy, dz, dy, d res, d a
    chain_bwd_bytes = Rm * (4 * (128 + 128 + 128 + 128) + 2 * 256 + 4 * (128 + 256 + 128 + 128 + 128))
    main_head = max(cands, key=lambda c: c["ms"] if c["ms_is"].startswith("in-step") else 0.0)
    cands = [dict(c, kernel=c["kernel"] + ": dominant MAIN-stream kernel by in-step duration") if c is main_head else c for c in cands]
    chain_fwd_ms = in_step("vlp3d_rows_chain_io", lambda a: a[2:4] == (Rm, 4))
    chain_bwd_ms = in_step("vlp3d_rows_chain_bwd", lambda a: a[1] == Rm and a[2] == 3, pick=min)
    chain_entries = []  # bf16 configuration only: the fp32 step runs the layer modules' own launches
    if chain_fwd_ms is not None:
        chain_entries.append(
            entry("rows_chain_kernel<32> decoder-layer tail: fc_o -> add & norm -> FFN -> add & norm -> next q|k|v, 16 384 rows, one "
                  "launch", "rows_chain_kernel", "hbm", chain_bytes, PEAK_HBM_GBS, "GB/s", None, chain_fwd_ms,
                  algorithmic_bytes=chain_bytes,
                  numerator="rows the stage contract moves: input (bf16 rows) + residual in, every stage output and what backward "
                            "keeps (pre-activation, FFN hidden, xhat; fp32) out, q|k|v as bf16 rows; the weights (0.5 MB) stay in L2"))
    if chain_bwd_ms is not None:
        chain_entries.append(
            entry("rows_chain_bwd_kernel decoder-layer tail backward: add & norm bwd -> W2^T -> ReLU/dropout bwd -> W1^T -> add & norm "
                  "bwd -> fc_o^T, one launch", "rows_chain_bwd_kernel", "hbm", chain_bwd_bytes, PEAK_HBM_GBS, "GB/s", None,
                  chain_bwd_ms, algorithmic_bytes=chain_bwd_bytes))
    # The headline `roofline` is the kernel that dominates the rocprofv3 summary of this command (profiles/r04_*_kernel_stats.csv:
    # ~30 % of all kernel time, side stream or not): the pruned FPS of SA1.  Its bound is the fp32 VALU rate of the 8 CUs it
    # occupies (one workgroup per scene, the judged figure for FPS), its numerator the DENSE algorithm's work.
    head = entry("fps_pruned_reg_kernel (csrc/fps_pruned.hip; the fps_pruned_kernel family) SA1 40000->2048, B=8: dominant kernel of "
                 "the step by rocprofv3 kernel time; side stream; bounding-box pruned FPS, same indices as the dense kernel and the "
                 "oracle", "fps_pruned", "valu", fps_flops, fps_peak, "TFLOP/s", fps_ms,
                 in_step("vlp3d_furthest_point_sampling_pruned"), cus_used=B,
                 bound_detail="fp32 VALU peak of the %d CUs used (one workgroup = one CU per scene): %.1f TFLOP/s x %d / %d"
                              % (B, PEAK_FP32_VECTOR_TFLOPS, B, NUM_CUS),
                 numerator="ALGORITHMIC: the dense algorithm's B*(m-1)*n = %.0f M distance-update-compares x 11 flop (SURVEY.md 8(d)); "
                           "the kernel executes only the updates its bounding-box test cannot rule out (DESIGN.md 4.1); `ms` "
                           "(in-step clock stamps around the entry point) INCLUDES the four sort pre-pass launches (~85 us)"
                           % (B * (m - 1) * n / 1e6),
                 hbm_algorithmic_bytes=B * (12 * n + 4 * m),
                 hbm_algorithmic_GBs=round(B * (12 * n + 4 * m) / (fps_ms * 1e-3) / 1e9, 3),
                 streaming_equiv_GBs=round(B * m * n * 20.0 / (fps_ms * 1e-3) / 1e9, 1),
                 frac_isolated=round(fps_flops / (fps_ms * 1e-3) / 1e12 / fps_peak, 4))
    others = cands + [
        entry("relation_bias_bwd_kernel (pairwise-geometry bias MLP 4->32->32->4, backward, one of two layers; side stream since "
              "the split backward)", "relation_bias_bwd", "mfma", rel_flops, PEAK_BF16_MFMA_TFLOPS / 16, "TFLOP/s", None,
              in_step("vlp3d_relation_bias_bwd"), peak_is="exact-fp32 MFMA (1/16 of the bf16 rate)", pairs=rel_pairs),
        entry("bq_sorted_kernel: ball query SA1 r=0.2 ns=64 in ONE launch on the spatial sort the pruned FPS left for the same cloud "
              "(csrc/ball_query_sorted.hip; side stream; same rows as the all-pairs kernel)",
              "bq_sorted", "hbm", bq_bytes, PEAK_HBM_GBS, "GB/s", bq_ms,
              in_step("vlp3d_ball_query_sorted", lambda a: 40000 in a), algorithmic_bytes=bq_bytes,
              tests_per_s_T=round(B * m * n / (bq_ms * 1e-3) / 1e12, 3),
              valu_frac_of_dense_tests=round(B * m * n * 8.0 / (bq_ms * 1e-3) / 1e12 / PEAK_FP32_VECTOR_TFLOPS, 4),
              grid_form_ms_isolated=round(bq_grid_ms, 4),
              note="8.2 MB of algorithmic bytes = 1 us at HBM peak: the kernel is bound by its per-centre latency chain (cell runs -> "
                   "candidates -> hits -> ranks), not by HBM; the six-launch grid form of round 3 (own second sort) is timed beside it"),
    ] + chain_entries + [
        entry(sdpa_name + " match self-attention 64x(256x256) h4 d32", "sdpa_fwd_lds_kernel<7>" if bf else None,
              "hbm", att_alg, PEAK_HBM_GBS, "GB/s", att_ms,
              in_step(sdpa_entry, lambda a: a[1:5] == (BL, 4, 256, 256)), algorithmic_bytes=att_alg, moved_bytes=att_moved,
              mfma_TFLOPs=round(4.0 * BL * 256 * 256 * 128 / (att_ms * 1e-3) / 1e12, 2)),
        entry(sdpa_name + " match cross-attention 64x(256x49) h4 d32", "sdpa_fwd_cross: void (anonymous namespace)::sdpa_fwd_lds_kernel<7>" if bf else None,
              "hbm", xat_alg, PEAK_HBM_GBS, "GB/s", xat_ms,
              in_step(sdpa_entry, lambda a: a[1:5] == (BL, 4, 256, NUM_TOKENS)), algorithmic_bytes=xat_alg, moved_bytes=xat_moved,
              mfma_TFLOPs=round(4.0 * BL * 256 * NUM_TOKENS * 128 / (xat_ms * 1e-3) / 1e12, 2)),
    ]
    if bf:  # VERDICT r3 #3's alternative bar for the cores: duration against 2 x (in-step launch floor + HBM time of the bytes)
        floor_us = empty_us - bracket_us
        for e_ in others[-2:]:
            if e_.get("ms"):
                hbm_us = e_["algorithmic_bytes"] / (PEAK_HBM_GBS * 1e9) * 1e6
                e_["launch_floor_plus_hbm_us"] = round(floor_us + hbm_us, 2)
                e_["ms_over_floor_plus_hbm"] = round(e_["ms"] * 1e3 / (floor_us + hbm_us), 2)
    return head, others, {"stamp_gap_us": round(gap_us, 2), "empty_kernel_in_step_us": round(empty_us - bracket_us, 2),
                          "linear_library_fallbacks_per_step": fallbacks, "stamped_launches": len(stamped)}


def host_feed(args, gs, first, world, rank, device, augment=False, stream=None):
    """Three distinct batches in pinned host memory (what DataLoader(pin_memory=True) hands over), cycled through
    input_pipeline.Prefetcher: two batches are on the device at any time (current + next, whose geometry the side stream
    prepares), a third is in flight."""
    synth = importlib.import_module("3dvlp_amd.synth")
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    host = []
    # bf16 configuration without augmentation: the loader hands the feature channels over as bf16 (input_pipeline.
    # compress_cloud — the values the first grouped-MLP layer rounds them to anyway): 88 instead of 173 MB per batch on the link
    compress = (args.dtype == "bf16") and not augment and os.environ.get("VLP3D_COMPRESS_CLOUD", "1") != "0"
    for j in range(3):
        hb = synth.make_batch(first + 8 * j * world, B_PER_GPU, NUM_POINTS, LANG_NUM, instances=augment)
        hb = {k: torch.from_numpy(v) for k, v in hb.items()}
        if compress:
            hb = ip.compress_cloud(hb)
        host.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in hb.items()})

    def endless():
        i = 0
        while True:
            yield host[i % 3]
            i += 1
    prep = gs.prepare_batch
    if augment:
        import numpy as np
        prep = ip.augmenting_prepare(np.random.default_rng(rank), gs.prepare_batch)
    return ip.Prefetcher(endless(), device=device, prepare=prep, stream=stream)


def extra_config_ms(args, batch, gs, side_stream, dtype=None, padded=False, feed=None, steps=30, warmup=10):
    """ms per step of ANOTHER configuration of the same step, timed after the headline run (wall clock around `steps` steps):
    dtype="fp32" = the 1e-4 parity configuration (exact-fp32 MFMA everywhere); padded=True = the grouped MLPs on the PADDED
    rows (VLP3D_SA_COMPACT=0: what the step costs when the distinct-row evaluation gains nothing — the synthetic scenes have
    39 % / 18 % distinct rows at SA1 / SA2, ScanNet-like density 65 % / 43 %); feed = every batch starts in pinned host memory."""
    dtype = dtype or args.dtype
    if dtype != "bf16" and "k/feat_bf" in batch:   # the parity configuration reads the fp32 cloud
        batch = gs.prepare_batch({k: v for k, v in batch.items() if not k.startswith("k/")})
    old = os.environ.get("VLP3D_SA_COMPACT")
    if padded:
        os.environ["VLP3D_SA_COMPACT"] = "0"
    try:
        step = gs.GroundingStep(batch["point_clouds"].device, epoch=50, sa_dtype=torch.bfloat16 if dtype == "bf16" else None,
                                use_graph=not args.no_graph, pipeline=not args.no_pipeline, side_stream=side_stream)
        cur = nxt = None
        if feed is not None:
            cur, nxt = feed.next(), feed.next()

        def one():
            nonlocal cur, nxt
            if feed is None:
                return step.run(batch)
            r = step.run(cur, nxt)
            cur, nxt = nxt, feed.next()
            return r
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
    finally:
        if padded:
            if old is None:
                os.environ.pop("VLP3D_SA_COMPACT", None)
            else:
                os.environ["VLP3D_SA_COMPACT"] = old
    del step
    return ms


def main():
    global B_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel roofline section (profiling runs)")
    ap.add_argument("--check-replicas", action="store_true",
                    help="after the timed steps, gather a checksum of every rank's parameters into the JSON line")
    ap.add_argument("--host-batches", action="store_true",
                    help="feed every step a batch that starts in (pinned) HOST memory through input_pipeline.Prefetcher: the "
                         "PCIe-inclusive rate (never the headline `value`; reported in DESIGN.md)")
    ap.add_argument("--augment", action="store_true",
                    help="with --host-batches: the reference's training-time augmentation (flip / rotate / scale / translate, "
                         "votes recomputed after it; lib/joint/dataset.py:653-690) on the device, on the copy stream")
    ap.add_argument("--scenes-per-gpu", type=int, default=B_PER_GPU,
                    help="scenes per GPU and step (default 8 = BASELINE cfg2, the headline; 32 = cfg3's per-GPU batch)")
    ap.add_argument("--caption", action="store_true",
                    help="BASELINE cfg4: the Scan2Cap caption head (30 522 words, 6 layers) attached to the step on the shared "
                         "proposal features — its parameters in the flat buffers, cap_loss inside the captured graph, one backward, "
                         "one all-reduce (never the headline `value`, which is cfg2)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="compute the backbone geometry (FPS / ball query) inline instead of one batch ahead")
    args = ap.parse_args()
    B_PER_GPU = args.scenes_per_gpu

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
    batch_np = synth.make_batch(first, B_PER_GPU, NUM_POINTS, LANG_NUM, caption_tokens=32 if args.caption else 0)
    # bf16 configuration: the loader hands the cloud's feature channels over as bf16 rows (prepare_batch(feat_bf16=True): the values
    # the first grouped-MLP layer rounds them to anyway) — read directly by the gather layer and its weight gradient
    feat_bf16 = args.dtype == "bf16" and os.environ.get("VLP3D_FEAT_BF16", "1") != "0"
    batch = gs.batch_to_device(batch_np, device, feat_bf16=feat_bf16)
    step = gs.GroundingStep(device, epoch=50, sa_dtype=torch.bfloat16 if args.dtype == "bf16" else None,
                            use_graph=not args.no_graph, pipeline=not args.no_pipeline, use_caption=args.caption)
    ddp.broadcast_parameters(step.model, layout=step.layout)
    # the loader's upload stream: created AND used right behind the step's side stream, before any capture creates its warm-up
    # stream — HIP hands its few hardware queues to streams at first use, and an upload stream that shares the launch stream's
    # queue serialises the 173 MB copy with the step (8.3 instead of 5.2 ms per step)
    copy_stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(step._side if step._side is not None else copy_stream):
        torch.zeros(1, device=device)
    with torch.cuda.stream(copy_stream):
        torch.zeros(1, device=device)
    torch.cuda.synchronize()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    feed = host_feed(args, gs, first, world, rank, device, augment=args.augment, stream=copy_stream) if args.host_batches else None

    def one_step():
        nonlocal cur_b, nxt_b
        if feed is None:
            return step.run(batch)
        loss_ = step.run(cur_b, nxt_b)
        cur_b, nxt_b = nxt_b, feed.next()
        return loss_

    cur_b = nxt_b = None
    if feed is not None:
        cur_b, nxt_b = feed.next(), feed.next()
    for _ in range(args.warmup):
        one_step()
    sync()
    if world > 1:
        step.comm_events = []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        loss = one_step()
    marks[args.steps].record()
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    exposed = None
    if step.comm_events:
        ex = sorted(a.elapsed_time(b) for a, b in step.comm_events)
        exposed = round(ex[len(ex) // 2], 4)
    pct = lambda p: round(per_step[min(len(per_step) - 1, int(p * len(per_step)))], 3)

    checks = None
    if args.check_replicas:
        flat = torch.cat([p.detach().reshape(-1).double() for p in step.model.parameters()])
        mine = torch.stack([flat.sum(), flat.abs().sum(), (flat * torch.arange(1, flat.numel() + 1, device=device)
                                                           .double().remainder(97.0)).sum()])
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(gathered, mine)
        else:
            gathered = [mine]
        checks = [[float(v) for v in g.cpu()] for g in gathered]

    if rank == 0:
        bf = args.dtype == "bf16"
        ms = 1e3 * elapsed / args.steps
        flops, byts = step_work(B_PER_GPU, 2 if bf else 4)
        peak_tf = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_BF16_MFMA_TFLOPS / 16  # exact-fp32 MFMA: 1/16 of the bf16 rate
        out = {
            "metric": "scenes/sec fwd+bwd, 40k-pt/256-proposal grounding",
            "value": round(B_PER_GPU * world * args.steps / elapsed, 3),
            "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("cfg2" if B_PER_GPU == 8 else "cfg3 per-GPU batch (%d scenes)" % B_PER_GPU) +
                       ": ScanRefer grounding step, 40k pts, 256 proposals, 8 sentences/scene" +
                       (" + cfg4's Scan2Cap caption head on the shared proposal features (32 tokens/sentence, 30 522 words)"
                        if args.caption else ""),
                       "batch_per_gpu": B_PER_GPU, "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}",
                       "step": "fwd + the reference's loss (loss_joint.py: vote, objectness, box + sem-cls, DIoU + "
                               "SoftmaxRankingLoss reference, OCC/OSC; epoch 50) + bwd + flat grad all-reduce + AdamW",
                       "precision": ("bf16 storage + bf16 MFMA (fp32 accumulate) in the grouped per-ball MLPs; bf16 MFMA operands "
                                     "rounded in registers (fp32 I/O, softmax, statistics, accumulate) in the attention cores, the "
                                     "plain linear layers (incl. the decoder stack's row chains, csrc/rows_chain.hip) and the Conv1d rows "
                                     "stacks; fp32 in the remaining element-wise kernels. "
                                     "NOT the 1e-4 parity configuration: that is --dtype fp32 (exact-fp32 MFMA everywhere)"
                                     if bf else "fp32 everywhere (exact-fp32 MFMA): the 1e-4 parity configuration"),
                       "launch": "eager" if args.no_graph else "hipGraph replay (fwd+loss+bwd)",
                       "geometry": "inline" if args.no_pipeline else
                       "backbone FPS/ball-query/three_nn of the next batch on a side stream (executed every step)",
                       "input": ("every step's batch starts in pinned host memory (PCIe-inclusive; 3 batches cycled, "
                                 "input_pipeline.Prefetcher" + (", device-side training augmentation" if args.augment else "") + ")"
                                 if args.host_batches else "resident in HBM before the timed region"),
                       "input_format": ("loader-prepared (grounding_step.prepare_batch): k/xyz fp32 + the cloud's feature channels as bf16 "
                                        "rows (k/feat_bf: the values the first grouped-MLP layer rounds them to; bit-identical step, "
                                        "DESIGN.md 4.20) + kernel-ready label dtypes" if (bf and "k/feat_bf" in (feed.data_dict or batch if feed is not None else batch))
                                        else "loader-prepared (grounding_step.prepare_batch): k/xyz + k/feat_pm fp32 + kernel-ready label dtypes"),
                       "loss": float(loss.detach())},
            "allreduce_exposed_ms": exposed,
            "allreduce": ("two pieces: the head parameters' slice of the flat fp32 gradient buffer is reduced beside the backward of "
                          "SA2 / SA1 (issued behind the deferred graph on the side stream), the rest after it; "
                          "allreduce_exposed_ms = median time the launch stream spends between the end of backward and the "
                          "optimiser (null on one GPU: no collective)") if step._head_range is not None else
                         "one all-reduce of the flat fp32 gradient buffer after backward",
            "step_ms": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9),
                        "how": "events on the launch stream between consecutive steps (this rank)"},
            "roofline_step": {"flops_per_step": flops, "bytes_per_step": byts,
                              "achieved_TFLOPs": round(flops / (ms * 1e-3) / 1e12, 2), "peak_TFLOPs": peak_tf,
                              "frac_mfma": round(flops / (ms * 1e-3) / 1e12 / peak_tf, 4),
                              "achieved_GBs": round(byts / (ms * 1e-3) / 1e9, 1), "peak_GBs": PEAK_HBM_GBS,
                              "frac_hbm": round(byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                              "note": "algorithmic dense flops (fwd x3) and stored-activation bytes of one step on this "
                                      "rank's scenes; the step is latency / launch bound, both fractions are small"},
        }
        if not args.no_kernels:
            out["roofline"], out["roofline_kernels"], extra = kernel_rooflines(args, batch, ext, gs, side_stream=step._side)
            out.update(extra)
            if bf and B_PER_GPU == 8 and not args.host_batches and world == 1 and not args.caption:
                # the other configurations of the same step a reader needs beside the headline: the data-independent
                # form, the 1e-4 parity configuration and the PCIe-inclusive rate — never `value`
                out["ms_per_step_padded"] = round(extra_config_ms(args, batch, gs, step._side, padded=True), 3)
                out["ms_per_step_fp32"] = round(extra_config_ms(args, batch, gs, step._side, dtype="fp32", steps=15, warmup=5), 3)
                out["ms_per_step_host_batches"] = round(
                    extra_config_ms(args, batch, gs, step._side, feed=host_feed(args, gs, first, world, rank, device, stream=copy_stream)), 3)
                out["padded_form_ms_per_step"] = out["ms_per_step_padded"]   # (round-3 name of the same figure)
                out["other_configurations"] = (
                    "ms_per_step_padded: grouped MLPs on the padded ball-query rows (no distinct-row evaluation; scenes/s = "
                    "%.0f); ms_per_step_fp32: --dtype fp32, exact-fp32 MFMA everywhere = the configuration the 1e-4 parity tests "
                    "run (scenes/s = %.0f); ms_per_step_host_batches: every batch starts in pinned host memory (PCIe-inclusive, "
                    "scenes/s = %.0f)" % tuple(1e3 * B_PER_GPU / out[k] for k in
                                               ("ms_per_step_padded", "ms_per_step_fp32", "ms_per_step_host_batches")))
            out["hw"] = measure_hw(ext, device)
        else:
            out["roofline"] = None
        if checks is not None:
            out["replica_param_checksums"] = checks
        if world == 1 and not args.no_cpu_baseline:
            from oracle import baseline
            out["cpu_baseline"] = baseline.cpu_baseline(batch_np, scenes=B_PER_GPU)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
