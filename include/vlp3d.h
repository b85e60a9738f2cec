/*
 * vlp3d.h — C ABI of libvlp3d_hip.so, the MI355X (gfx950) implementation of the
 * 3DVLP point-cloud + language grounding hot path.
 *
 * Drop-in boundary: these entry points are what the reference's pybind module
 * `pointnet2._ext` (lib/pointnet2/_ext_src/src/bindings.cpp:11-24) binds, plus the
 * fused ops that replace Python-level hot loops.  Conventions (all entry points):
 *   - plain device pointers + extents, `stream` is a hipStream_t passed as void*
 *     (NULL = the null stream); the call only ENQUEUES work, it never synchronises;
 *   - tensors are dense row-major ("contiguous") fp32 / int32 exactly as the
 *     reference requires (include/utils.h:15-30);
 *   - the CALLER owns every buffer, including scratch; the library allocates nothing
 *     and keeps no mutable global state (re-entrant from any host thread);
 *   - return value: 0 = enqueued; VLP3D_EINVAL (-22) = bad argument (nothing
 *     enqueued); > 0 = a hipError_t from the launch.  The library never exits or
 *     throws (the reference's CUDA_CHECK_ERRORS exit(-1)s: cuda_utils.h:35-44).
 *   - outputs are fully written by the kernels; callers need not zero-fill them
 *     (the reference's host wrappers torch::zeros() them) EXCEPT where noted
 *     ("accumulates": the *_grad scatter ops zero the buffer themselves too).
 *
 * Index outputs (FPS / ball query / three_nn) are bit-exact with the reference's
 * CUDA kernels under the fp32 evaluation order documented at vlp3d_fp_contract().
 */
#ifndef VLP3D_H
#define VLP3D_H

#ifdef __cplusplus
extern "C" {
#endif

#define VLP3D_OK 0
#define VLP3D_EINVAL (-22)

/* ABI version, bumped on any signature change. */
int vlp3d_abi_version(void);

/* Which fp32 contraction the distance expressions were built with:
 *   0: ((a*a)+(b*b))+(c*c)   1: fma(c,c, fma(a,a, b*b)) [default: what NVPTX emits under
 *   nvcc's default -fmad=true for the reference's expression]   2: fma(c,c, fma(b,b, a*a)). */
int vlp3d_fp_contract(void);

/* ---- the nine pointnet2._ext ops -------------------------------------------------- */

/* replaces furthest_point_sampling — sampling.cpp:70-91 / sampling_gpu.cu:74-234.
 * xyz (B,N,3) f32; temp (B,N) f32 scratch (contents ignored on entry, unspecified on exit);
 * idx (B,m) i32 out.  idx[:,0] = 0.  Same skip rule (|p|^2 <= 1e-3) and the same
 * tie order as the reference's 512-thread LDS tree. */
int vlp3d_furthest_point_sampling(const float *xyz, int B, int N, int m, float *temp, int *idx, void *stream);

/* FPS of a point set that is itself an earlier FPS's output in sampling order (every backbone level after the first,
 * backbone_module.py:93-117: sa2..sa4 sample from the previous level's new_xyz): vlp3d_fps_prefix_check proves in two
 * parallel kernels that the sequential result is 0..m-1 (strict inequalities at every step, no skipped point) and
 * writes *not_prefix = 0, or 1 when the proof fails; vlp3d_furthest_point_sampling_cond then fills idx with 0..m-1 or
 * runs the sequential kernel.  Same output as vlp3d_furthest_point_sampling either way.  v: (B, m) floats of scratch;
 * N <= 65536.  not_prefix == NULL in _cond: unconditional. */
int vlp3d_fps_prefix_check(const float *xyz, int B, int N, int m, float *v, int *not_prefix, void *stream);
int vlp3d_furthest_point_sampling_cond(const float *xyz, int B, int N, int m, float *temp, int *idx,
                                       const int *not_prefix, void *stream);

/* Same output as vlp3d_furthest_point_sampling, computed with distance-bound pruning (csrc/fps_pruned.hip):
 * points are Morton-sorted into 64-point slots and a slot whose bounding box is farther from the new sample than
 * its current maximum is skipped.  workspace: vlp3d_fps_workspace_bytes(B,N) bytes of scratch.  N <= 131072. */
long long vlp3d_fps_workspace_bytes(int B, int N);
int vlp3d_furthest_point_sampling_pruned(const float *xyz, int B, int N, int m, void *workspace,
                                         long long workspace_bytes, int *idx, void *stream);

/* replaces gather_points — sampling.cpp:20-43. points (B,C,N), idx (B,M) -> out (B,C,M). */
int vlp3d_gather_points(const float *points, const int *idx, int B, int C, int N, int M, float *out, void *stream);

/* replaces gather_points_grad — sampling.cpp:45-69. grad_out (B,C,M) -> grad_points (B,C,N)
 * (zeroed by the call, then scatter-added). */
int vlp3d_gather_points_grad(const float *grad_out, const int *idx, int B, int C, int N, int M, float *grad_points,
                             void *stream);

/* replaces ball_query — ball_query.cpp:13-37 / ball_query_gpu.cu:14-59.
 * new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample) i32: first nsample indices k (ascending)
 * with d2 < radius*radius, padded with the first hit; all-zero row when there is none. */
int vlp3d_ball_query(const float *new_xyz, const float *xyz, int B, int N, int M, float radius, int nsample,
                     int *idx, void *stream);

/* Same output as vlp3d_ball_query, bit for bit, through a uniform grid (csrc/ball_query_grid.hip): cell edge >= radius,
 * only the 27 cells around a centre are tested, the hits are ranked by index and the nsample smallest written in
 * ascending order with the reference's padding.  workspace: vlp3d_ball_query_grid_workspace_bytes(B,N) bytes, 16-byte
 * aligned; five kernels + one memset on `stream`. */
long long vlp3d_ball_query_grid_workspace_bytes(int B, int N);
int vlp3d_ball_query_grid(const float *new_xyz, const float *xyz, int B, int N, int M, float radius, int nsample,
                          void *workspace, long long workspace_bytes, int *idx, void *stream);

/* replaces group_points — group_points.cpp:17-40. points (B,C,N), idx (B,M,S) -> out (B,C,M,S). */
int vlp3d_group_points(const float *points, const int *idx, int B, int C, int N, int M, int S, float *out,
                       void *stream);

/* replaces group_points_grad — group_points.cpp:42-65. grad_out (B,C,M,S) -> grad_points (B,C,N)
 * (zeroed by the call, then scatter-added). */
int vlp3d_group_points_grad(const float *grad_out, const int *idx, int B, int C, int N, int M, int S,
                            float *grad_points, void *stream);

/* replaces three_nn — interpolate.cpp:19-45. unknown (B,n,3), known (B,m,3) ->
 * dist2 (B,n,3) f32 SQUARED distances ascending, idx (B,n,3) i32. */
int vlp3d_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2, int *idx,
                   void *stream);

/* replaces three_interpolate — interpolate.cpp:47-75. points (B,C,m), idx (B,n,3), weight (B,n,3)
 * -> out (B,C,n). */
int vlp3d_three_interpolate(const float *points, const int *idx, const float *weight, int B, int C, int m, int n,
                            float *out, void *stream);

/* replaces three_interpolate_grad — interpolate.cpp:77-104 — with the TRUE adjoint that
 * interpolate_gpu.cu:121-148 intends (the reference host function calls the forward wrapper by
 * mistake, interpolate.cpp:95; DESIGN.md "known deviations").  grad_out (B,C,n) -> grad_points (B,C,m)
 * (zeroed by the call, then scatter-added). */
int vlp3d_three_interpolate_grad(const float *grad_out, const int *idx, const float *weight, int B, int C, int n,
                                 int m, float *grad_points, void *stream);

/* ---- fused ops replacing Python-level hot loops ------------------------------------ */

/* replaces utils/nn_distance.py:32-59 without the (B,N,M,C) temporaries.
 * pc1 (B,N,3), pc2 (B,M,3); mode 0 = squared L2, 1 = L1, 2 = Huber(delta).
 * -> dist1 (B,N) f32, idx1 (B,N) i64, dist2 (B,M) f32, idx2 (B,M) i64 (first minimum wins). */
int vlp3d_nn_distance(const float *pc1, const float *pc2, int B, int N, int M, int mode, float delta, float *dist1,
                      long long *idx1, float *dist2, long long *idx2, void *stream);

/* replaces QueryAndGroup's gather stage (lib/pointnet2/pointnet2_utils.py:343-355: group(xyz), -= centre,
 * /= radius, group(features), cat) with GEMM-ready rows.  xyz (B,N,3), new_xyz (B,M,3), idx (B,M,S) i32,
 * feat_pm (B,N,C) f32 POINT-MAJOR features, C % 4 == 0.
 * -> out (B*M*S, C+4): [features(C) | (xyz[idx]-new_xyz)/radius (3) | 0], f32 or bf16 (out_bf16 != 0).
 * Pass radius = 1 for normalize_xyz=False. */
int vlp3d_group_rows(const float *xyz, const float *new_xyz, const int *idx, const float *feat_pm, int B, int N,
                     int M, int S, int C, float radius, void *out, int out_bf16, void *stream);

/* adjoint of vlp3d_group_rows: dout (B*M*S, C+4) f32/bf16 -> dfeat_pm (B,N,C), dxyz (B,N,3), dnew_xyz (B,M,3),
 * each optional (NULL = not needed); zeroed by the call, then scatter-added with contiguous float atomics. */
int vlp3d_group_rows_grad(const void *dout, int dout_bf16, const int *idx, int B, int N, int M, int S, int C,
                          float radius, float *dfeat_pm, float *dxyz, float *dnew_xyz, void *stream);

/* ---- grouped per-ball MLP of a set-abstraction layer on the matrix cores (csrc/sa_mlp.hip) ------------
 * Replaces, for PointnetSAModuleVotes.forward (lib/pointnet2/pointnet2_modules.py:233-267) and its autograd
 * backward: group_points x2, sub/div, cat, 3 x (1x1 conv, BatchNorm2d, ReLU), max_pool2d over nsample.
 * Rows r = (b*M + m)*S + s; Y_l (R x cout_l) pre-activations, G_l masked gradients; bf16_io: 0 = fp32 storage +
 * exact-fp32 MFMA, 1 = bf16 storage + bf16 MFMA (fp32 accumulate).  Per-channel vectors are fp32, batch
 * statistics fp64.  R = B*M*S must be a multiple of 32; cout in {32,64,128,256}.
 *
 * COMPACT ROW MAP (optional; bf16_io only; csrc/sa_compact.hip).  ball_query pads a ball with copies of its first
 * neighbour (ball_query_gpu.cu:14-49) and the reference runs the padded tensor through the whole stack.  With
 * (crow, rowptr, nballs) the stack is evaluated on the DISTINCT rows of every ball, stored back to back — same result up
 * to the order of float additions: a row's multiplicity w enters the BatchNorm batch sums and the BatchNorm-backward
 * term (dY summed over the copies = k1 (G - w (k2 + yhat k3))), everything downstream is linear in that sum.
 *   vlp3d_sa_compact(idx (B,M,S), ...) -> rowptr (B*M + 1 ints; rowptr[B*M] = number of compact rows P, read on the
 *   device — launches stay sized for the dense worst case R) and crow (R x int4: global point row b*N + idx, (ball << 8) |
 *   position in the ball, float bits of w, 0; rows P .. roundup32(P)-1 are zero-weight dummies).
 * Every entry below that takes (crow, rowptr, nballs) treats its (R x .) matrices as compact when crow != NULL;
 * NULL = dense rows.  Matrices keep their R-row allocation; rows past roundup32(P) are never touched. */
int vlp3d_sa_compact(const int *idx, int B, int N, int M, int S, int *rowptr, void *crow, void *stream);

/* layer 1: Y = [feat_pm[idx] | (xyz[idx]-new_xyz)/radius | 0] * W^T; W (cout x K) in column order
 * [features(C) | xyz(3) | 0], K >= C+4, K % 8 (fp32) / 16 (bf16) == 0.
 * stats: (vlp3d_sa_stat_slabs(R) x 2 x cout) f64 — one [sum | sumsq] slab per workgroup, fully written (no atomics;
 * vlp3d_sa_bn_fold sums the slabs).  The same convention holds for `tstats` of vlp3d_sa_bwd_layer. */
/* bf16_io of vlp3d_sa_fwd_gather and of vlp3d_sa_wgrad (gather != 0): bit 0 = bf16 storage / MFMA as everywhere; bit 1 (with
 * bit 0; K <= 160, or cout = 128 with K <= 288 and C % 8 == 0; M*S % 32 == 0 or a compact row map) = feat_pm points at BF16
 * feature rows of (C + 7) & ~7 columns, columns [C, ..) zero — the loader's bf16 copy of the cloud's channels
 * (input_pipeline.compress_cloud / prepare_batch) or the previous level's vlp3d_sa_pool_rows output: half the gathered bytes, the
 * same numbers (the fp32 rows are rounded to bf16 on their way into LDS anyway). */
int vlp3d_sa_stat_slabs(long long R);
int vlp3d_sa_fwd_gather(const float *xyz, const float *new_xyz, const int *idx, const float *feat_pm, int B, int N,
                        int M, int S, int C, float radius, const void *W, int K, int cout, void *Y, double *stats,
                        int bf16_io, const void *crow, const int *rowptr, int nballs, void *stream);
/* layers 2..: Y = relu(Yin*scale + shift) * W^T, Yin (R x K), W (cout x K). */
int vlp3d_sa_fwd_layer(const void *Yin, long long R, int K, const float *scale, const float *shift, const void *W,
                       int cout, void *Y, double *stats, int bf16_io, const void *crow, const int *rowptr, int nballs,
                       void *stream);
/* out (BM x C) f32 = relu(sel*scale + shift), sel = max_s Y (scale >= 0) / min_s Y (scale < 0); sel_idx u8
 * (position inside the ball; with rowptr != NULL inside the ball's compact rows). */
int vlp3d_sa_pool(const void *Y, long long BM, int S, int C, const float *scale, const float *shift, float *out,
                  unsigned char *sel_idx, int bf16_io, const int *rowptr, void *stream);
/* vlp3d_sa_pool that also writes the pooled rows as bf16 (out_bf16 (BM x C), C % 8 == 0): the next level's gather layer and its
 * weight gradient read them through bf16_io bit 1 instead of rounding the fp32 rows themselves (the same values). */
int vlp3d_sa_pool_rows(const void *Y, long long BM, int S, int C, const float *scale, const float *shift, float *out,
                       void *out_bf16, unsigned char *sel_idx, int bf16_io, const int *rowptr, void *stream);
/* G (BM*S x C) = dP routed to the selected sample where out > 0 (max-pool + ReLU backward). */
int vlp3d_sa_pool_grad(const float *dP, const float *out, const unsigned char *sel_idx, long long BM, int S, int C,
                       void *G, int bf16_io, void *stream);
/* G_{l-1} = relu-mask(BNbwd(G_l, Y_l) * W_l), tstats (2 x kprev) f64 += [sum g, sum g*yhat] of layer l-1.
 * bn5 = [rstd | -mean*rstd | gamma*rstd | mean(g) | mean(g*yhat)] (5 x ld) of layer l; WT = W_l^T (kprev x ld);
 * prev4 = [scale | shift | rstd | -mean*rstd] (4 x kprev) of layer l-1.
 * LAST layer: pass G = NULL and the pooled tensors (pool_g = gsel of vlp3d_sa_pool_tstats, sel (BM x ld) u8, pool_S = nsample):
 * the masked gradient is then synthesised on the fly (no dense (R x ld) gradient matrix is ever written). */
int vlp3d_sa_bwd_layer(const void *G, const void *Y, long long R, int ld, const float *bn5, const void *WT, int kprev,
                       const void *Yprev, const float *prev4, void *Gprev, double *tstats, const float *pool_g,
                       const unsigned char *pool_sel, int pool_S, int bf16_io, const void *crow, const int *rowptr,
                       int nballs, void *stream);
/* layer 1 input gradient, scatter-added (NOT zeroed here) into dfeat_pm (B,N,C) / dxyz (B,N,3) / dnew_xyz (B,M,3)
 * (each optional).  WT = W_1^T zero-padded to (kpad x ld), kpad % 32 == 0. */
int vlp3d_sa_bwd_gather(const void *G, const void *Y, int ld, const float *bn5, const void *WT, int kpad,
                        const int *idx, int B, int N, int M, int S, int C, float radius, float *dfeat_pm, float *dxyz,
                        float *dnew_xyz, int bf16_io, const void *crow, const int *rowptr, int nballs, void *stream);
/* dW (cout x K) f32 = sum_r BNbwd(G,Y)[r]^T A[r]; A = relu(Yprev*scale+shift) (gather == 0) or the gathered
 * layer-1 rows (gather != 0).  partials: scratch of max_blocks*cout*K floats (one slab per workgroup, summed by
 * a second kernel: no contended atomics); dW is fully written.  G = NULL + pooled tensors as in vlp3d_sa_bwd_layer. */
int vlp3d_sa_wgrad(const void *G, const void *Y, long long R, int cout, const float *bn5, int gather,
                   const void *Yprev, int K, const float *scale, const float *shift, const float *xyz,
                   const float *new_xyz, const int *idx, const float *feat_pm, int N, int M, int S, int C,
                   float radius, float *dW, float *partials, int max_blocks, const float *pool_g,
                   const unsigned char *pool_sel, int pool_S, int bf16_io, int defer_reduce, const void *crow,
                   const int *rowptr, int nballs, void *stream);

/* per-channel bookkeeping of the fused layer (one launch each instead of ~20 framework kernels):
 * bn_fold: vec (4 x C) = [scale | shift | rstd | -mean*rstd] from the fp64 batch sums `stats` (training) or the
 * running statistics (eval); training also updates running_mean/var (may be NULL) like nn.BatchNorm. */
int vlp3d_sa_bn_fold(const double *stats, int nslab, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, int C, long long R, float eps, float momentum, int training, float *vec,
                     void *stream);
/* bn5 (5 x C) backward constants, dgamma, dbeta (C) from vec, gamma and the reductions t (2 x C) f64. */
int vlp3d_sa_bn_bwd_consts(const float *vec, const float *gamma, const double *t, int nslab, int C, long long R,
                           int training, float *bn5, float *dgamma, float *dbeta, void *stream);
/* t (vlp3d_sa_pool_tstats_slabs(BM) x 2 x C) f64 = per-workgroup [sum g, sum g*yhat] of the LAST layer computed from the
 * pooled tensors (slabs for vlp3d_sa_bn_bwd_consts's nslab: nothing to clear, the sums do not depend on an atomic order);
 * gsel (BM x C) f32 = dP where out > 0 else 0 (the ReLU-masked pooled gradient the last-layer loaders read). */
int vlp3d_sa_pool_tstats_slabs(long long BM);
int vlp3d_sa_pool_tstats(const float *dP, const float *out, const float *gamma, const float *beta, long long BM, int C,
                         double *t, float *gsel, void *stream);

/* Weight layouts of one fused SA stack in one launch.  W1 (c0, C+3) [xyz | features] as the reference's first
 * conv stores it, W2 (c1, c0), W3 (c2, c1), all fp32.  out (bf16 or fp32 by bf16_io) receives, back to back:
 * W1p (c0,K1) = [features | xyz | 0], W2 (c1,c0), W3 (c2,c1), W1p^T zero-padded (kpad,c0), W2^T (c0,c1), W3^T (c1,c2). */
int vlp3d_sa_prep_weights(const float *W1, const float *W2, const float *W3, int C, int c0, int c1, int c2, int K1,
                          int kpad, void *out, int bf16_io, void *stream);

/* ---- plain linear layers on the grouped-MLP kernels (fp32, exact-fp32 MFMA) -----------------------------------
 * replace nn.Linear forward / weight-gradient of the attention projections, FFNs and heads on this path
 * (models/transformer/attention.py:22-25, mmattention.py:40-41, match_module.py:30-40): Y = X W^T + bias.
 * R % 32 == 0, K % 8 == 0, N in {32,64,128,160,256,288} (fwd) / {64,128,256} (wgrad).
 * vlp3d_linear_dgrad: dX (R,K) = dY (R,N) W with W (N,K) as nn.Linear stores it (no transposed copy); K % 32 == 0.
 * vlp3d_linear_wgrad: dW (N,K) = dY^T X; with_bias != 0 appends the bias gradient: dW then has N*K + N floats and
 * `partials` max_blocks * (N*K + N). */
/* bf16_mma != 0 (the timing configuration, together with the bf16 grouped MLPs): operands are rounded to bf16 in
 * registers / LDS and contracted with v_mfma_f32_32x32x16_bf16; memory I/O and accumulation stay fp32 (K % 16 == 0,
 * R <= 65536 for fwd; otherwise the exact-fp32 form runs). */
int vlp3d_linear_fwd(const float *X, const float *W, const float *bias, long long R, int K, int N, float *Y,
                     int bf16_mma, void *stream);
int vlp3d_linear_dgrad(const float *dY, const float *W, long long R, int N, int K, float *dX, const float *base, int bf16_mma,
                       void *stream); /* base (R,K, optional, bf16_mma only): dX = dY W + base — the gradient that arrives
                                         through a residual connection beside the layer, added in the epilogue */
int vlp3d_linear_wgrad(const float *dY, const float *X, long long R, int K, int N, float *dW, float *partials,
                       int max_blocks, int with_bias, int defer_reduce, int bf16_mma, void *stream);

/* Deferred slab sums.  The three weight-gradient entries (vlp3d_sa_wgrad, vlp3d_linear_wgrad, vlp3d_rows_wgrad) write
 * per-workgroup partial results ("slabs") and then sum them with a second launch.  With defer_reduce != 0 they stop
 * after the slabs (dW / dbias are not touched and may be NULL) and the caller sums the slabs of MANY launches at once:
 *   nblk   = ceil(T / ceil(T / max_blocks)), T = R / 32 — the number of slabs the launch wrote
 *   slab   = [n_mat = N*K matrix floats | n_bias column sums (0 or N)]
 *   dst    : element (row, k) of the matrix goes to dst[row*ldo + k]; with ncol_out > 0 only k < ncol_out is kept and
 *            stored at column (k + rot) % ncol_out (vlp3d_sa_wgrad, gather layer: slab columns are [features | xyz | pad],
 *            the parameter is [xyz | features]: ncol_out = C + 3, rot = 3)
 * descs is a HOST array, consumed during the call. */
typedef struct {
  const float *partials;
  float *dst;
  float *dbias;
  int nblk, n_mat, n_bias, K, ldo, ncol_out, rot;
} vlp3d_slab_reduce_desc;
int vlp3d_slab_reduce_batch(const vlp3d_slab_reduce_desc *descs, int count, void *stream);

/* ---- the training loss of the grounding step (csrc/joint_loss.hip) ------------------------------------------
 * replaces lib/loss_helper/loss_joint.py:26-227 (get_joint_loss with detection + reference, run.sh:1) and what it
 * calls: loss_detection.py:24-258 (vote, objectness, box + sem-cls on recover_assigned_gt_bboxes),
 * loss_grounding.py:129-365 (compute_diou_loss: utils/box_util.py:488-529 DIoU, best-IoU / smooth labels,
 * lib/loss_helper/loss.py:6-17 SoftmaxRankingLoss) — minus their Python loops and host syncs.
 * Differentiable inputs: vote_xyz (B,S,3), obj_scores (B,K,2), heading_scores / heading_res_norm (B,K,NH), rois (B,K,6),
 * sem_scores (B,K,NC), agg_xyz / pred_center / pred_size (B,K,3), cluster_ref (B*L,K).
 * Labels: seed_xyz (B,S,3), seed_inds (B,S) i32 into N, vote_label (B,N,9), vote_mask (B,N) f32, center_label (B,G,3),
 * heading_class_label i32 / heading_residual_label f32 / size_class_label i32 / sem_cls_label i32 (B,G),
 * size_residual_label (B,G,3), ref_center / ref_size (B,L,3) (decoded GT boxes of the sentences), lang_num (B) i32,
 * coin: device float, the IoUs are gated by the objectness arg-max when coin < 0.5 (the train-time
 * `data_dict["random"] < 0.5`; pass 1.0 otherwise), mean_size (num_size_cluster,3).
 * smooth_labels: 1 = epoch < 50 label smoothing (0.95 / 0.05 split), 0 = hard labels.
 * fwd: part = vlp3d_joint_loss_rows(B,S,K,L) rows of 16 doubles (scratch), sums = 16 doubles (kept for bwd),
 *      out[15] = vote, objectness, heading_cls, heading_reg, size_distance, sem_cls, box, ref, diou,
 *                total = 10*(vote + 0.1*objectness + box) + w_ref*ref + w_diou*diou, pos_ratio, neg_ratio, obj_acc,
 *                max_iou_rate_0.25, max_iou_rate_0.5;
 *      assign (B,K) i32 = object_assignment; objlab (B,K) i32 = objectness_label | objectness_mask << 1;
 *      rowinfo (B,L,4) i32 = per sentence: valid (best IoU >= 0.25), argmax IoU, argmax gated IoU, #gated IoU >= 0.25.
 * bwd: gout = device scalar d/d(out[9]) (NULL = 1); every gradient buffer is fully written. */
long long vlp3d_joint_loss_rows(int B, int S, int K, int L);
int vlp3d_joint_loss_fwd(const float *vote_xyz, const float *obj_scores, const float *heading_scores,
                         const float *heading_res_norm, const float *rois, const float *sem_scores, const float *agg_xyz,
                         const float *pred_center, const float *pred_size, const float *cluster_ref, const float *seed_xyz,
                         const int *seed_inds, const float *vote_label, const float *vote_mask, const float *center_label,
                         const int *heading_class_label, const float *heading_residual_label, const int *size_class_label,
                         const float *size_residual_label, const int *sem_cls_label, const float *ref_center,
                         const float *ref_size, const int *lang_num, const float *coin, const float *mean_size, int B, int S,
                         int N, int K, int G, int L, int NH, int NC, float near_thr, float far_thr, float w0, float w1,
                         float w_ref, float w_diou, int smooth_labels, double *part,
                         double *sums, float *out, int *assign, int *objlab, int *rowinfo, void *stream);
int vlp3d_joint_loss_bwd(const float *vote_xyz, const float *obj_scores, const float *heading_scores,
                         const float *heading_res_norm, const float *rois, const float *sem_scores, const float *agg_xyz,
                         const float *pred_center, const float *pred_size, const float *cluster_ref, const float *seed_xyz,
                         const int *seed_inds, const float *vote_label, const float *vote_mask, const float *center_label,
                         const int *heading_class_label, const float *heading_residual_label, const int *size_class_label,
                         const float *size_residual_label, const int *sem_cls_label, const float *ref_center,
                         const float *ref_size, const int *lang_num, const float *coin, const float *mean_size, int B, int S,
                         int N, int K, int G, int L, int NH, int NC, float near_thr, float far_thr, float w0, float w1,
                         float w_ref, float w_diou, int smooth_labels,
                         const double *sums, const int *assign, const int *objlab, const int *rowinfo, const float *gout,
                         float *d_vote_xyz, float *d_obj_scores, float *d_heading_scores, float *d_heading_res_norm,
                         float *d_rois, float *d_sem_scores, float *d_agg_xyz, float *d_pred_center, float *d_pred_size,
                         float *d_cluster_ref, void *stream);

/* ---- OCC / OSC InfoNCE of the contrast module (csrc/contrast.hip) ------------------------------------------
 * Core of models/constrast_module/constrast_module.py:53-131 for all (scene, sentence) pairs at once.
 * text (B,L,D), box (B,K,D), boxi (B,K,D): L2-normalised text_proj / pc_proj / pc_proj_iou outputs; obj (B,K) f32 in
 * {0,1} (objectness argmax); gt_center, gt_size (B,L,3) (size grown by 1e-2 inside, as the reference does);
 * pred_center, pred_size (B,K,3); lang_num (B) int64.  K <= 1024, L <= 64, D % 4 == 0.
 * fwd -> out2 = [lang_con_loss (OCC), iou_con_loss (OSC)], lse (B, L+K) (kept for backward).
 * bwd: g_occ / g_osc = device scalars d(loss)/d(out2[0]) / d(out2[1]) (NULL = 0) -> dtext (B,L,D), dbox (B,K,D),
 * dboxi (B,K,D), all fully written; dS (B, L+K, K) f32 scratch. */
int vlp3d_contrast_fwd(const float *text, const float *box, const float *boxi, const float *obj,
                       const float *gt_center, const float *gt_size, const float *pred_center, const float *pred_size,
                       const long long *lang_num, int B, int L, int K, int D, float *out2, float *lse, void *stream);
int vlp3d_contrast_bwd(const float *text, const float *box, const float *boxi, const float *obj,
                       const float *gt_center, const float *gt_size, const float *pred_center, const float *pred_size,
                       const long long *lang_num, int B, int L, int K, int D, const float *lse, const float *g_occ,
                       const float *g_osc, float *dS, float *dtext, float *dbox, float *dboxi, void *stream);

/* ---- add & norm of the attention / FFN blocks (csrc/add_norm.hip) ------------------------------------------
 * out = LayerNorm(x + dropout(y)) as in models/transformer/attention.py:128-130 and mmattention.py:84-86, one
 * kernel forward, one + a slab sum backward.  x, y, out, xhat: (R, D) f32, D in {64,128,256}; rstd: (R).
 * p = dropout probability (0: none / eval).  The mask is a counter-based hash of (*seed, call_id, element index),
 * recomputed in backward: pass the same seed word and call_id to both; advance *seed between steps.
 * mask (R,D) u8 or NULL: optional copy of the keep mask (tests).
 * bwd: dx = d(out)/d(x), dy = dx * mask/(1-p); dgamma_dbeta (2,D) = [sum dout*xhat | sum dout];
 * partials: vlp3d_add_norm_blocks(R) * 2 * D floats of scratch. */
int vlp3d_add_norm_blocks(long long R);
int vlp3d_add_norm_fwd(const float *x, const float *y, const float *gamma, const float *beta, long long R, int D,
                       float p, const unsigned long long *seed, int call_id, float eps, float *out, float *xhat,
                       float *rstd, unsigned char *mask, void *stream);
int vlp3d_add_norm_bwd(const float *dout, const float *xhat, const float *rstd, const float *gamma, long long R, int D,
                       float p, const unsigned long long *seed, int call_id, float *dx, float *dy, float *partials,
                       float *dgamma_dbeta, void *stream);

/* add & norm over REPLICATED rows: x, y hold R / rep rows; every group of `seq` source rows feeds `rep` consecutive output
 * groups (the proposals tiled over the sentences, match_module.py:127) with independent dropout masks per output element.
 * vlp3d_rep_sum2: the adjoint of the replication for two gradients at once: sa[g,k,:] = sum_l a[g,l,k,:] (rows_out =
 * R / rep rows out). */
int vlp3d_add_norm_rep_fwd(const float *x, const float *y, const float *gamma, const float *beta, long long R, int D, int rep,
                           int seq, float p, const unsigned long long *seed, int call_id, float eps, float *out, float *xhat,
                           float *rstd, void *stream);
int vlp3d_rep_sum2(const float *a, const float *b, long long rows_out, int D, int rep, int seq, float *sa, float *sb,
                   void *stream);

/* Pre-norm residual stream of the caption decoder (models/caption_module/transformer_captioner.py:132-145
 * SublayerConnection: x + dropout(sublayer(norm(x))); :117-129 its own LayerNorm = a*(x-mean)/(std+eps)+b with the
 * unbiased std): one launch gives the new stream value sum_out = x + dropout_p(y) (y NULL: = x) AND out = norm(sum_out).
 * std_mode 1 = that LayerNorm, 0 = nn.LayerNorm.  kappa (R): scratch kept for backward (required with std_mode 1, else
 * optional).  bwd: dout = gradient of out, dres = gradient of sum_out or NULL; dx = total gradient of x,
 * dy = dx * mask/(1-p) (NULL without y); defer_reduce 1 leaves the vlp3d_add_norm_blocks(R) slabs of 2*D floats in
 * `partials` for vlp3d_slab_reduce_batch instead of forming dgamma_dbeta.  Other arguments as vlp3d_add_norm_*. */
int vlp3d_sum_norm_fwd(const float *x, const float *y, const float *gamma, const float *beta, long long R, int D,
                       float p, const unsigned long long *seed, int call_id, float eps, int std_mode, float *sum_out,
                       float *out, float *xhat, float *rstd, float *kappa, unsigned char *mask, void *stream);
int vlp3d_sum_norm_bwd(const float *dout, const float *dres, const float *xhat, const float *rstd, const float *kappa,
                       const float *gamma, long long R, int D, float p, const unsigned long long *seed, int call_id,
                       float *dx, float *dy, float *partials, float *dgamma_dbeta, int defer_reduce, void *stream);

/* ---- box decode of the proposal module (csrc/box_decode.hip) ------------------------------------------------
 * Replaces decode_pred_box (models/proposal_module/proposal_module_fcos.py:94-144) + get_3d_box_batch
 * (utils/box_util.py:361-385) for n = B*num_proposal proposals, NH heading bins:
 *   vote_xyz (n,3), heading_scores (n,NH), heading_residuals (n,NH), rois (n,6: distances to the 6 faces)
 *   -> heading (n), size (n,3), centre (n,3), corners (n,8,3), heading_class (n) int32 (kept for backward).
 * bwd: d_heading (n) / d_size (n,3) / d_centre (n,3), any of them NULL = zero
 *   -> d_rois (n,6), d_residuals (n,NH), d_vote_xyz (n,3), all fully written.  corners carry no gradient
 *   (the reference builds them from .detach().cpu().numpy() values). */
int vlp3d_box_decode_fwd(const float *vote_xyz, const float *heading_scores, const float *heading_residuals,
                         const float *rois, int n, int NH, float *heading, float *size, float *centre, float *corners,
                         int *heading_class, void *stream);
int vlp3d_box_decode_bwd(const float *rois, const float *heading, const int *heading_class, const float *d_heading,
                         const float *d_size, const float *d_centre, int n, int NH, float *d_rois, float *d_residuals,
                         float *d_vote_xyz, void *stream);

/* ---- pairwise-geometry attention bias of the relation module (csrc/relation_bias.hip) ------------------
 * Replaces models/proposal_module/relation_module.py:72-92 per layer: out[b,c,i,j] = MLP([c_j - c_i, |c_j - c_i|])[c]
 * with MLP = Linear(4,32) ReLU LayerNorm(32) Linear(32,32) ReLU LayerNorm(32) Linear(32,4) (:26-37).
 * params: vlp3d_relation_bias_nparam() floats = [W1(32x4) b1 g1 be1 W2(32x32) b2 g2 be2 W3(4x32) b3]
 * (the order of nn.Sequential.parameters()).  centre (B,K,3) -> out (B,4,K,K). */
int vlp3d_relation_bias_nparam(void);
int vlp3d_relation_bias_fwd(const float *centre, const float *params, int B, int K, float *out, void *stream);
/* dout (B,4,K,K) -> dparams (nparam, fully written); slabs: scratch of nblocks*nparam floats (one partial per workgroup).
 * bf16_mma != 0: the 32 x 32 layer products and the parameter-gradient updates as bf16 MFMA (operands rounded in registers,
 * fp32 accumulation; the step's timing configuration); 0: exact fp32 MFMA (parity configuration). */
int vlp3d_relation_bias_bwd(const float *centre, const float *params, const float *dout, int B, int K,
                            float *dparams, float *slabs, int nblocks, int bf16_mma, void *stream);

/* replaces the att = softmax(QK^T/sqrt(dk) [+bias | *w] [mask]) V core of
 * models/transformer/attention.py:63-75 without materialising att.
 * q (B,nq,H*D), k/v (B,nk,H*D) are the OUTPUTS of fc_q/fc_k/fc_v (head h = columns h*D..h*D+D-1);
 * bias: (B,H,nq,nk) f32 or NULL (bias_mode 0 none, 1 add, 2 mul); mask: (B,nk) f32 or NULL,
 * broadcast over heads and queries (0 -> masked_fill(-10000)).  D must be 32.
 * -> out (B,nq,H*D) f32 (the input of fc_o), lse (B,H,nq) f32 (log-sum-exp, kept for backward).
 * bf16_mma: 0 = exact-fp32 MFMA (within fp32 round-off of the reference: the 1e-4 parity configuration);
 *           1 = operands rounded to bf16 in registers, fp32 accumulation and softmax (timing configuration). */
int vlp3d_sdpa_fwd(const float *q, const float *k, const float *v, const float *bias, int bias_mode,
                   const float *mask, int B, int H, int nq, int nk, int D, float *out, float *lse, int bf16_mma,
                   int ldq, int ldk, int ldv, void *stream);

/* backward of vlp3d_sdpa_fwd: dout (B,nq,H*D) -> dq, dk, dv (shapes and ROW STRIDES of q,k,v; fully written),
 * dbias (B,H,nq,nk) or NULL.  delta (B,H,nq) f32 scratch.
 * ldq / ldk / ldv: row strides in floats of q / k / v (>= H*D, multiple of 4; H*D = contiguous) — the three may be
 * column blocks of one merged projection output, and dq / dk / dv column blocks of one merged gradient. */
int vlp3d_sdpa_bwd(const float *q, const float *k, const float *v, const float *bias, int bias_mode,
                   const float *mask, const float *out, const float *lse, const float *dout, int B, int H, int nq,
                   int nk, int D, float *dq, float *dk, float *dv, float *dbias, float *delta, int bf16_mma,
                   int ldq, int ldk, int ldv, void *stream);

/* The same cores on operands that already ARE bf16 in memory — SURVEY.md §8(d)'s bytes for the attention cores (q, k, v, out
 * touched once as bf16).  io bits: 1 = q, 2 = k and v, 4 = out (written by _fwd_io, read by _bwd_io) hold bf16 rows instead of
 * fp32; strides stay in ELEMENTS (a bf16 operand needs stride % 8 == 0); dout and dq / dk / dv stay fp32 with the strides of
 * q / k / v.  Built: io = 0, 5 (proposal <-> token cross-attention, mmattention.py:75-80: the 16 384-row q and out as bf16,
 * k / v from the small fp32 projection of the tokens), 7 (self-attention on a merged bf16 q|k|v, mmattention.py:70-73).
 * io != 0 requires bf16_mma, the LDS kernels' shapes (nk <= 288, nq <= 512, B*H >= 64) and bias_mode 0 in backward; else
 * VLP3D_EINVAL.  The bf16-MFMA kernels round q / k / v to bf16 anyway: the forward results are the fp32-row results to the
 * bit (out rounded once more when stored), the backward differs only through delta = rowsum(dout * out). */
int vlp3d_sdpa_fwd_io(const void *q, const void *k, const void *v, const float *bias, int bias_mode, const float *mask,
                      int B, int H, int nq, int nk, int D, void *out, float *lse, int bf16_mma, int ldq, int ldk, int ldv,
                      int io, void *stream);
int vlp3d_sdpa_bwd_io(const void *q, const void *k, const void *v, const float *bias, int bias_mode, const float *mask,
                      const void *out, const float *lse, const float *dout, int B, int H, int nq, int nk, int D, float *dq,
                      float *dk, float *dv, float *dbias, float *delta, int bf16_mma, int ldq, int ldk, int ldv, int io,
                      void *stream);

/* ---- exact-fp32 dense stacks on point-major rows (csrc/rows_mlp.hip) -------------------------------------------
 * The 1x1 Conv(+bias) -> BatchNorm -> ReLU chains of PointnetFPModule (pointnet2_modules.py:403-416), VotingModule
 * (voting_module.py:33-60) and StandardROIHeads (roi_heads.py:15-147) as products on ROW-MAJOR matrices with the
 * BatchNorm / ReLU stages folded into the neighbouring products (same scheme as vlp3d_sa_*; per-channel vectors come
 * from vlp3d_sa_bn_fold / vlp3d_sa_bn_bwd_consts).  R % 32 == 0, R <= 2^24.
 * rows_fwd:   Y (R x N, stride ldy) = A W^T [+ bias]; A = X or relu(X*scale + shift) with a_vec = [scale | shift | ..]
 *             (length K each); stats != NULL (then bias must be NULL): per-column sums of Y into
 *             [vlp3d_rows_slabs(R)][2][N] fp64 slabs.  W (N x K) row-major, K % 32 == 0, N % 32 == 0.
 * rows_dgrad: dA (R x K) = dY W with dY = G or BatchNorm-backward(G, Ypre; bn5 [5][N]); p_vec != NULL masks with the
 *             ReLU of the previous layer (Yprev, p_vec = its vec [4][K]) and writes the two BatchNorm-backward column
 *             sums to tstats slabs [vlp3d_rows_slabs(R)][2][K].  K % 32 == 0, N % 32 == 0.
 * rows_wgrad: dW[:, 0:K] (stride ldo) = dY^T A (+ dbias = column sums of G when bn5 == NULL); see csrc/sa_mlp.hip.
 * rows_act / rows_act_bwd: the activation of the last BatchNorm layer and its backward (+ column sums, nslab =
 *             vlp3d_rows_act_slabs(R)).  slope == NULL: ReLU; else PReLU with that per-channel slope (C floats), and
 *             dslope_slabs [nslab][C] fp64 receives the partial sums of its gradient.
 * fp_rows / fp_rows_grad: X = [three_interpolate(known) | unknown] on point-major features (pointnet2_modules.py:393-411,
 *             blend order of interpolate_gpu.cu:103-104) and the adjoint w.r.t. known (m <= 1024). */
/* One job of vlp3d_rows_wgrad_batch: the operands of vlp3d_rows_wgrad (dW / ldo / dbias are not needed: the batch always
 * leaves the slabs in `partials`; with_bias asks for the column sums of G in the slabs). */
typedef struct vlp3d_rows_wgrad_job {
  const float *G;
  const float *Ypre;
  int ldg;
  const float *bn5;
  const float *X;
  int lda;
  const float *a_scale, *a_shift;
  long long R;
  int K, N;
  float *partials;
  int max_blocks, with_bias;
  int x_bf16;  /* plain-linear jobs only (bn5 == a_scale == NULL): X holds bf16 rows, lda in elements */
} vlp3d_rows_wgrad_job;
/* Weight gradients of `count` rows-stack layers (Conv1d + BatchNorm1d + ReLU stacks of the FP / voting / ROI / relation heads,
 * pointnet2_modules.py:371-416, voting_module.py:33-60, roi_heads.py:15-147) in a few launches: jobs of one kernel
 * instantiation share a launch (bf16-MFMA configuration).  Slabs as vlp3d_rows_wgrad(..., defer_reduce = 1) writes them. */
int vlp3d_rows_wgrad_batch(const vlp3d_rows_wgrad_job *jobs, int count, void *stream);

/* vlp3d_linear_fwd (bf16_mma = 1) with the result stored as bf16 rows (R x N): the query projection in front of an attention
 * core that reads bf16 rows (vlp3d_sdpa_fwd_io).  R % 32 == 0, K % 16 == 0, N % 64 == 0. */
int vlp3d_linear_fwd_rows16(const float *X, const float *W, const float *bias, long long R, int K, int N, void *Y16,
                            void *stream);

/* One weight-gradient job of vlp3d_linear_wgrad_batch: the arguments of vlp3d_linear_wgrad(dY, X, R, K, N, NULL, partials,
 * max_blocks, with_bias, defer_reduce = 1, bf16_mma = 1). */
typedef struct vlp3d_linear_wgrad_job {
  const float *dY;
  const float *X;
  float *partials;
  long long R;
  int K, N, max_blocks, with_bias;
  int x_bf16;  /* X holds bf16 rows (R x K): an attention core's bf16 output as the layer's input (vlp3d_sdpa_fwd_io) */
} vlp3d_linear_wgrad_job;
/* The weight gradients of `count` plain linear layers (nn.Linear backward, attention.py / mmattention.py / match_module.py
 * projections and FFNs) in one or a few launches instead of one each: they feed nothing but the optimiser, so the step driver
 * queues them during backward and runs them together (bf16-MFMA configuration; N % 64 == 0, N <= 512, K <= 256).  Every job
 * writes its own slabs exactly as the single entry would; sum them with vlp3d_slab_reduce_batch. */
int vlp3d_linear_wgrad_batch(const vlp3d_linear_wgrad_job *jobs, int count, void *stream);

/* bf16_mma != 0: bf16 MFMA operands rounded in registers / LDS (timing configuration), fp32 I/O, statistics, accumulation */
int vlp3d_rows_slabs(long long R);
int vlp3d_rows_fwd(const float *X, int ldx, long long R, int K, const float *a_vec, const float *W, const float *bias, int N,
                   float *Y, int ldy, double *stats, int bf16_mma, void *stream);
/* vlp3d_rows_fwd with the weight K-major: WT (K x ldw), WT[k][n] = W[n][k] (a copy made by vlp3d_transpose_batch) */
int vlp3d_rows_fwd_wt(const float *X, int ldx, long long R, int K, const float *a_vec, const float *WT, int ldw,
                      const float *bias, int N, float *Y, int ldy, double *stats, int bf16_mma, void *stream);
int vlp3d_rows_dgrad(const float *G, const float *Ypre, int ldg, const float *bn5, const float *W, long long R, int N, int K,
                     const float *Yprev, int ldprev, const float *p_vec, float *dA, int lda, double *tstats, int bf16_mma,
                     void *stream);
int vlp3d_rows_wgrad(const float *G, const float *Ypre, int ldg, const float *bn5, const float *X, int lda,
                     const float *a_scale, const float *a_shift, long long R, int K, int N, float *dW, int ldo, float *dbias,
                     float *partials, int max_blocks, int defer_reduce, int bf16_mma, void *stream);
int vlp3d_rows_act(const float *Y, long long R, int C, const float *vec, const float *slope, float *out, void *stream);
int vlp3d_rows_act_slabs(long long R);
int vlp3d_rows_act_bwd(const float *dOut, const float *Y, long long R, int C, const float *vec, const float *slope, float *G,
                       double *tstats, double *dslope_slabs, void *stream);
int vlp3d_fp_rows(const float *known, const float *unknown, const int *idx, const float *weight, int B, int n, int m, int C1,
                  int C2, float *X, void *stream);
int vlp3d_fp_rows_grad(const float *dX, const int *idx, const float *weight, int B, int n, int m, int C1, int ld,
                       float *d_known, void *stream);
/* the same adjoint without LDS atomics, through the inverse of the three_nn map: inv_start (B*m + 1), inv_refs (B*n*3) =
 * vlp3d_sa_inverse(idx viewed as (B, n, 3), crow NULL, N = m) — references are flat (b*n + p)*3 + k into `weight` */
int vlp3d_fp_rows_grad_csr(const float *dX, const float *weight, const int *inv_start, const int *inv_refs, int B, int m,
                           int C1, int ld, float *d_known, void *stream);

/* ---- fused glue between the matrix-core kernels (csrc/glue.hip) ---------------------------------------------
 * roi_split: out (R x ld) = [heading_reg NH | heading_cls NH | box 6 | objectness 2 | sem NC] of the merged ROI
 *   predictors (roi_heads.py:135-147) -> contiguous heading_residuals_normalized, heading_residuals (x res_scale),
 *   heading_scores, rois = exp(box), objectness_scores, sem_cls_scores, objectness arg-max and sem arg-max (int64).
 *   roi_split_bwd gathers the six gradient pieces (NULL = zero) into d(out), zeroing the unused columns.
 * vote_epilogue: vote_xyz = seed_xyz + net[:, 0:3], vote_features = normalise(seed_f + net[:, 3:3+C]) (voting_module.py:
 *   51-58 with vote_factor 1, jointnet.py:148-149); norm (R) is kept for the backward, which writes d(seed_f) and
 *   d(net) (R x ld, fully).
 * l2norm_rows(_bwd): F.normalize(x, dim=-1, eps) on (R x C) rows and its backward.
 * relation_inputs: relation_module.py:95-122 — obj_feat (B,K,128) rows of the point cloud's multiview block with the
 *   reference's indexing, manual_bbox_feat (B,K,27) = [box centre | corners - centre], centre (B,K,3) = corner mean.
 * copy_paste_map: match_module.py:97-121 as an index map src (B*K) i32 (identity when coin >= 0.5); gather_rows /
 *   scatter_rows_add: out[i] = x[src[i]] and its adjoint (dx zeroed by the caller).  B*K <= 8192. */
int vlp3d_roi_split(const float *out, int ld, long long R, int NH, int NC, float res_scale, float *hreg, float *hres,
                    float *hcls, float *rois, float *obj, float *sem, long long *obj_mask, long long *sem_arg, void *stream);
int vlp3d_roi_split_bwd(const float *d_hreg, const float *d_hres, const float *d_hcls, const float *d_rois, const float *d_obj,
                        const float *d_sem, const float *rois, long long R, int NH, int NC, float res_scale, float *d_out,
                        int ld, void *stream);
int vlp3d_vote_epilogue(const float *seed_xyz, const float *seed_f, const float *net, int ld, long long R, int C,
                        float *vote_xyz, float *vote_f, float *norm, void *stream);
int vlp3d_vote_epilogue_bwd(const float *d_vote_xyz, const float *d_vote_f, const float *vote_f, const float *norm, long long R,
                            int C, float *d_seed_f, float *d_net, int ld, void *stream);
int vlp3d_l2norm_rows(const float *x, long long R, int C, float eps, float *y, float *norm, void *stream);
int vlp3d_l2norm_rows_bwd(const float *g, const float *y, const float *norm, long long R, int C, float eps, float *dx,
                          void *stream);
/* pc (B,N,Cpc) point-major rows whose columns col0 .. col0+127 are the 128 multiview channels: the raw cloud (Cpc = 3+132,
 * col0 = 6) or the loader's feature split k/feat_pm (Cpc = 132, col0 = 3). */
int vlp3d_relation_inputs(const float *pc, int Cpc, int col0, int N, const int *seed_inds, int S, const int *vote_inds,
                          const float *corners, int B, int K, float *obj_feat, float *bbox_feat, float *centre, void *stream);
/* the same from bf16 rows (the loader's bf16 copy of the feature channels, Cpc = its row stride): values widened */
int vlp3d_relation_inputs_bf16(const void *pc, int Cpc, int col0, int N, const int *seed_inds, int S, const int *vote_inds,
                               const float *corners, int B, int K, float *obj_feat, float *bbox_feat, float *centre,
                               void *stream);
int vlp3d_copy_paste_map(const long long *obj_mask, int B, int K, const float *coin, int *src, void *stream);
int vlp3d_gather_rows(const float *x, const int *src, long long R, int D, float *out, void *stream);
int vlp3d_scatter_rows_add(const float *g, const int *src, long long R, int D, float *dx, void *stream);

/* AdamW (torch.optim.AdamW semantics) over one flat fp32 parameter / gradient / moment buffer; active (n) u8: 0 = the
 * element's parameter received no gradient this step and is left untouched.  bias_c1 = 1 - beta1^t, sqrt_bias_c2 =
 * sqrt(1 - beta2^t) for step t (host-computed). */
int vlp3d_adamw_flat(float *p, const float *g, float *m, float *v, const unsigned char *active, long long n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float bias_c1, float sqrt_bias_c2, void *stream);

/* ---- hardware-denominator probes (csrc/hwprobe.hip; measurement only) --------------------
 * vlp3d_probe_read: streaming 16-byte-load read of `bytes` (multiple of 16, >= 16 KiB) with `blocks` workgroups;
 * vlp3d_probe_mfma_bf16: blocks*4 waves each issue iters*4 independent v_mfma_f32_32x32x16_bf16
 *   (flops = blocks*4*iters*4*2*32*32*16);  vlp3d_probe_fma_f32: blocks*256 lanes x iters x 8 independent FMAs
 *   (flops = blocks*256*iters*16).  `sink`: one float, never written for real data. */
int vlp3d_probe_read(const void *buf, long long bytes, int blocks, float *sink, void *stream);
int vlp3d_probe_mfma_bf16(int iters, int blocks, float *sink, void *stream);
int vlp3d_probe_fma_f32(int iters, int blocks, float *sink, void *stream);
/* vlp3d_stamp: one thread writes the 100 MHz device clock to *slot (in-stream time stamps bracketing a kernel inside a
 * captured step); vlp3d_probe_empty: a kernel that does nothing on a (blocks, threads) grid — the launch floor. */
/* vlp3d_fps_pruned_profile: the ROUND-3 pruned kernel + per-phase shader-clock counts of its
 * iteration (phases: (B, 8) u64, entries 0..4: slot test | slot updates | wave candidate | LDS + barrier | block reduction). */
int vlp3d_fps_pruned_profile(const float *xyz, int B, int N, int m, void *workspace, long long workspace_bytes, int *idx,
                             unsigned long long *phases, void *stream);
/* Backward of the LAST layer of a grouped MLP without its pre-activation (csrc/sa_last.hip; bf16 storage; (C2, C3) one of
 * (64,128), (128,256), (128,128): vlp3d_sa_last_supported).  With a2 = relu(bn(Y2)) the layer's input and y3 = a2 W3^T, the
 * BatchNorm backward dY3 = k1 G - w (alpha + beta y3) (G: the pooled gradient at the arg-max sample of each ball) gives
 * dA2 = (k1 G) W3 - w (W3^T alpha + a2 W3^T diag(beta) W3) and dW3 = (k1 G)^T a2 - alpha (x) sum w a2 - diag(beta) W3 sum w a2^T a2:
 * neither reads Y3.  vlp3d_sa_last_dgrad == vlp3d_sa_bwd_layer(G = NULL, pool_g, pool_sel): G2 (rows x C2) bf16 masked gradient
 * of the layer below + nslab [sum g | sum g yhat] slabs (unused ones zeroed); vlp3d_sa_last_wgrad == vlp3d_sa_wgrad(...pool...)
 * with defer_reduce: `blocks` slabs (C3 x C2) fp32 in `partials`, to be summed.  gsel / sel: vlp3d_sa_pool_tstats / vlp3d_sa_pool;
 * vec2: vlp3d_sa_bn_fold of the layer below; bn5: vlp3d_sa_bn_bwd_consts of the last layer; W3T (C2 x C3) / W3 (C3 x C2) bf16:
 * vlp3d_sa_prep_weights; crow / rowptr / nballs: the compact row map or NULL / NULL / 0 (dense rows r = bm * S + s). */
int vlp3d_sa_last_supported(int C2, int C3);
int vlp3d_sa_last_dgrad(const void *Y2, const float *vec2, const float *bn5, const void *W3T, const float *gsel,
                        const unsigned char *sel, long long BM, int S, int C2, int C3, void *G2, double *tstats, int nslab,
                        const void *crow, const int *rowptr, int nballs, void *stream);
int vlp3d_sa_last_wgrad(const void *Y2, const float *vec2, const float *bn5, const void *W3, const float *gsel,
                        const unsigned char *sel, long long BM, int S, int C2, int C3, float *partials, int blocks,
                        const void *crow, const int *rowptr, int nballs, void *stream);
/* vlp3d_ball_query_sorted: ball_query (same output as vlp3d_ball_query, ball_query_gpu.cu:14-59) in ONE launch on the spatial
 * sort that vlp3d_furthest_point_sampling_pruned has just left in `fps_workspace` for the SAME xyz (same B, N; the workspace
 * not overwritten since): the backbone's first level samples its centres from the cloud it then queries.  xyz is read only by
 * the per-centre fall-back (more than 2048 candidates or 1024 hits in one neighbourhood). */
int vlp3d_ball_query_sorted(const float *new_xyz, const float *xyz, int B, int N, int M, float radius, int nsample,
                            const void *fps_workspace, long long fps_workspace_bytes, int *idx, void *stream);
/* vlp3d_fps_pruned_trace: diagnostic form.  variant 0 = the register-resident kernel (round 4, N <= 65536), 1 = the round-3
 * kernel (running minima in LDS / L2); lds_slots >= 0 overrides the number of slots per wave whose points live in LDS
 * (0..12 / 0..9); with `phases` (any non-null pointer for variant 0) the profiling instantiation runs; with `trace` too (u32,
 * B*m*16*8 words) lane 0 of every wave writes per iteration j: trace[((b*m + j)*16 + wave)*8 + k] = cycles of the five
 * phases, then LDS-resident / L2-resident slots updated and slots reduced again (tools/fps_trace.py). */
int vlp3d_fps_pruned_trace(const float *xyz, int B, int N, int m, void *workspace, long long workspace_bytes, int *idx,
                           unsigned long long *phases, unsigned *trace, int lds_slots, int variant, void *stream);
int vlp3d_stamp(unsigned long long *slot, void *stream);
int vlp3d_probe_empty(int blocks, int threads, int *sink, void *stream);

/* data_dict's reporting tensors from vlp3d_joint_loss_fwd's compact outputs, in the reference's dtypes: object_assignment
 * i64 (B,K), objectness_label i64 (B,K), objectness_mask f32 (B,K) (loss_detection.py:101-108), cluster_labels f32 (B,L,K)
 * (hard one-hot of the best-IoU proposal, zero rows below IoU 0.25; loss_grounding.py:84-92). */
int vlp3d_joint_loss_report(const int *assign, const int *objlab, const int *rowinfo, int B, int K, int L,
                            long long *assign64, long long *label64, float *mask, float *cluster_labels, void *stream);

/* dropout_p(act(z)) and its backward in one element-wise launch each (attention.py:104-112: dropout(relu(.)) of the
 * feed-forward block; match_module.py:40-47: Dropout(GELU(.))).  kind 0 = ReLU, 1 = GELU (erf).  The keep mask is the
 * add & norm hash of (seed word, call_id, element) — never stored; `mask` (optional, n bytes) receives it for tests.
 * dout == NULL: out = forward value; dout != NULL: out = dz.  n % 4 == 0, n < 2^32. */
int vlp3d_act_dropout(const float *z, const float *dout, long long n, int kind, float p, const unsigned long long *seed,
                      int call_id, float *out, unsigned char *mask, void *stream);

/* Small linear layers outside the MFMA kernels' shapes (csrc/glue.hip):
 * smallk: out (R,N) = base (R,N, optional) + x[:, :K] W^T + b,  K <= 32, N in {64,128,256}  (relation_module.py:64
 *         bbox_embedding = Linear(27,128)); backward writes ceil(R/64) slabs [N x 32 | N] — sum them with
 *         vlp3d_slab_reduce_batch {n_mat = 32 N, K = 32, ldo = ncol_out = K, n_bias = N}.
 * rowdot: y (R) = x (R,K) w + b  (match_module.py:47 Linear(128,1)); backward writes dx (optional) and
 *         ceil(R/rows_per_block) slabs [dw (K) | db,0,0,0] — vlp3d_slab_reduce_batch {n_mat = K = ldo = K + 4}. */
int vlp3d_smallk_fwd(const float *x, int ldx, const float *W, const float *bias, const float *base, long long R, int K, int N,
                     float *out, void *stream);
int vlp3d_smallk_bwd(const float *dy, const float *x, int ldx, long long R, int K, int N, float *slabs, void *stream);
int vlp3d_rowdot_fwd(const float *x, const float *w, const float *b, long long R, int K, float *y, void *stream);
int vlp3d_rowdot_bwd(const float *dy, const float *x, const float *w, long long R, int K, int rows_per_block, float *dx,
                     float *slabs, void *stream);

/* count device-to-device copies in one launch (descs: HOST array, consumed during the call; overlapping entries are
 * the caller's problem).  Replaces torch._foreach_copy_ for the hand-over of the prepared backbone geometry. */
typedef struct {
  const void *src;
  void *dst;
  long long bytes;
} vlp3d_copy_desc;
int vlp3d_copy_batch(const vlp3d_copy_desc *descs, int count, void *stream);

/* dst (cols x ld_dst floats) = src (rows x cols, row-major, contiguous)^T for `count` matrices in one launch.  Used by the
 * rows stacks (3dvlp_amd/row_mlp.py prepared_weights) to give the forward product of a Conv1d / nn.Linear layer the K-major
 * weight image that vlp3d_rows_fwd_wt reads with coalesced fragment loads (reference: the cuBLAS GEMMs behind
 * lib/pointnet2/pytorch_utils.py:49-90 pick their own operand layout; the parameters keep the reference's (N x K) storage). */
typedef struct vlp3d_transpose_desc {
  const void *src;
  void *dst;
  int rows, cols, ld_dst;
} vlp3d_transpose_desc;
int vlp3d_transpose_batch(const vlp3d_transpose_desc *descs, int count, void *stream);

/* A chain of nn.Linear stages over 64-row tiles that stay in LDS between the stages (csrc/rows_chain.hip) — the row-local
 * part of the reference's decoder layer between two attention cores, one launch instead of one per module:
 *   attention.py:75 fc_o -> :128-130 LayerNorm(q + dropout(out))                      [-> the next block's fc_q / fc_q|k|v]
 *   mmattention.py:36-50 FFN linear1 -> ReLU -> dropout -> linear2 -> :84-86 LayerNorm(dropout(f) + x)
 *   match_module.py:40-47 Linear -> GELU -> Dropout (x2)
 * Stage s maps the current tile t_s (R x K) to t_{s+1}:
 *   v = t_s W^T + bias                                   W (N x K) fp32 row-major; N, K multiples of 128, K <= 256, N <= 384
 *   v_out != NULL: v is stored (R x N)
 *   act_kind >= 0: v = dropout_{act_p}(act(v)), act 0 = ReLU, 1 = GELU (erf); mask = the add & norm hash of (seed, act_call,
 *                  row * N + column) — the mask vlp3d_act_dropout draws for the same call id; h_out != NULL: stored
 *   has_ln: N = 128; v = LayerNorm_{gamma,beta,eps}(res + dropout_{ln_p}(v)) with res (R x 128) and the mask of
 *           vlp3d_add_norm_fwd for call id ln_call; ln_out, xhat (R x 128) and rstd (R) are stored (all required)
 *   t_{s+1} = v (bf16 in LDS: the next stage's MFMA operand; the last stage may have N = 384, the others N <= 256).
 * X (R x K0) fp32 contiguous, K0 = stage 0's K.  bf16 MFMA operands, fp32 accumulation: the timing configuration — results
 * equal the unfused bf16 entry points (vlp3d_linear_fwd bf16_mma=1, vlp3d_add_norm_fwd, vlp3d_act_dropout) up to the
 * summation order of the LayerNorm statistics.  Backward uses the unfused entries on the stored tensors. */
#define VLP3D_CHAIN_MAX_STAGES 6
typedef struct vlp3d_chain_stage {
  const float *W, *bias;
  int N, K;
  float *v_out;
  int act_kind;
  float act_p;
  int act_call;
  float *h_out;
  int has_ln;
  const float *res, *gamma, *beta;
  float ln_p;
  int ln_call;
  float eps;
  float *ln_out, *xhat, *rstd;
  int v_out_bf16;  /* plain stage (no act, no LayerNorm): v_out receives bf16 rows (R x N) instead of fp32 */
  int h_out_bf16;  /* activation stage: h_out receives bf16 rows — the values the next stage and the weight gradient multiply */
} vlp3d_chain_stage;
int vlp3d_rows_chain(const float *X, long long R, const vlp3d_chain_stage *stages, int nstages,
                     const unsigned long long *seed, void *stream);
/* The same launch with X given as bf16 rows (x_bf16 != 0: R x K0 bf16, e.g. vlp3d_sdpa_fwd_io's output with io bit 4): the
 * tile goes to LDS without a conversion — the values the fp32 form would have rounded to. */
int vlp3d_rows_chain_io(const void *X, int x_bf16, long long R, const vlp3d_chain_stage *stages, int nstages,
                        const unsigned long long *seed, void *stream);

/* Backward of such a chain's row-local part, again one launch over 32-row tiles (csrc/rows_chain.hip).  The gradient walks
 * `ngemm` input-gradient products with `ngemm + 1` POINTS between them; point 0 sees the incoming gradient G (R x gemms[0].K),
 * point j > 0 the result of product j-1 (R x gemms[j-1].N).  At a point, in this order:
 *   + base   (R x N, e.g. the gradient a residual connection delivers from outside the chain), + the kept residual gradient
 *   op 0: nothing.
 *   op 1: add & norm backward (vlp3d_sum_norm_bwd's arithmetic, N = 128) with aux = xhat, rstd, gamma, dropout (p, call):
 *         the gradient of the residual input goes to dres_out and / or is kept for a later point's add_kept; the gradient of
 *         the normalised branch continues; part (blocks x 2 x 128, blocks = vlp3d_rows_chain_bwd_blocks(R)) receives this
 *         workgroup's [sum dout*xhat | sum dout] for vlp3d_slab_reduce_batch.
 *   op 2: activation + dropout backward (vlp3d_act_dropout's arithmetic) with aux = z, act_kind, (p, call).
 *   g_out != NULL: the gradient after the op is stored (R x N) — what the layer's weight gradient reads; required at the last point.
 * Product j: out = g Wt^T with Wt (N x K) row-major = the forward weight TRANSPOSED (dX = dY W needs W's columns as rows; the
 * K-major copies row_mlp.PreparedWeights keeps serve both directions).  N, K multiples of 128, <= 256. */
typedef struct vlp3d_chain_bwd_point {
  const float *base;
  int add_kept;
  int op;
  const float *aux, *rstd, *gamma;
  float p;
  int call;
  int act_kind;
  float *g_out, *dres_out;
  int keep;
  float *part;
  int aux_bf16;  /* op 2, ReLU only: aux = the stage's OUTPUT h = dropout(relu(z)) as bf16 rows (vlp3d_chain_stage.h_out_bf16)
                  * instead of z: the forward pass then need not store the pre-activation at all (v_out NULL) */
} vlp3d_chain_bwd_point;
typedef struct vlp3d_chain_bwd_gemm {
  const float *Wt;
  int N, K;
} vlp3d_chain_bwd_gemm;
int vlp3d_rows_chain_bwd_blocks(long long R);
int vlp3d_rows_chain_bwd(const float *G, long long R, const vlp3d_chain_bwd_point *points, const vlp3d_chain_bwd_gemm *gemms,
                         int ngemm, const unsigned long long *seed, void *stream);

/* Caption head (csrc/caption.hip).  Replaces, for `TransformerDecoderModel(30522)` of models/jointnet/jointnet.py:104:
 *
 * cap_attn — `attention()` + the head split/merge of `MultiHeadedAttention.forward`
 *   (models/caption_module/transformer_captioner.py:32-42, 66-78): softmax(q k^T / 4, masked_fill(mask == 0, -1e9)) ->
 *   dropout_p -> v for d_k = 16, T <= 64 positions.  qkv: (n*T, ld) rows [q | k | v] of H*16 columns each (the merged
 *   projection's output, ld >= 3*H*16, ld % 4 == 0); kmask (n, T) bytes, 1 = the key may be attended (NULL = all);
 *   causal 1 adds j <= i.  out (n*T, H*16), lse (n*H, T) scratch kept for backward; dqkv (n*T, 3*H*16) contiguous.
 *   Dropout mask: the add & norm hash of (seed word, call_id, ((seq*H + h)*T + i)*T + j), never stored.
 *
 * vocab_ce — `Generator.forward` (:106-114, log_softmax(proj(x))) fused with the token cross entropy and arg-max of
 *   lib/loss_helper/loss_captioning.py:37-71: X (R, 128), W (V, 128), bias (V) -> lse (R), nll (R) = lse - logit[target],
 *   argmax (R); the (R, V) logits never reach memory.  partials: vlp3d_vocab_ce_partial_bytes(R, V) bytes of scratch.
 *   bwd: coef (R) = dLoss/d nll; dX (R, 128) (zeroed inside), dW (V, 128), dbias (V); any of the three may be NULL
 *   (dbias only together with dW).  bf16_mma 1: operands rounded to bf16 (fp32 accumulate, softmax); 0: exact fp32. */
int vlp3d_cap_attn_fwd(const float *qkv, int ld, const unsigned char *kmask, int n, int T, int H, int causal, float p,
                       const unsigned long long *seed, int call_id, float *out, float *lse, void *stream);
int vlp3d_cap_attn_bwd(const float *qkv, int ld, const unsigned char *kmask, int n, int T, int H, int causal, float p,
                       const unsigned long long *seed, int call_id, const float *out, const float *lse, const float *dout,
                       float *dqkv, void *stream);
int vlp3d_vocab_ce_splits(long long R, int V);
long long vlp3d_vocab_ce_partial_bytes(long long R, int V);
int vlp3d_vocab_ce_fwd(const float *X, const float *W, const float *bias, const int *target, long long R, int V, int bf16_mma,
                       void *partials, float *lse, float *nll, int *argmax, void *stream);
int vlp3d_vocab_ce_bwd(const float *X, const float *W, const float *bias, const int *target, const float *lse,
                       const float *coef, long long R, int V, int bf16_mma, float *dX, float *dW, float *dbias, void *stream);

/* Answer loss of the joint QA + grounding step (lib/loss_helper/loss_answering.py:11-13):
 * out[0] = sum(binary_cross_entropy_with_logits(x, t)) / rows for (rows, cols) logits x and soft targets t;
 * partial: vlp3d_bce_logits_blocks(rows*cols) doubles of scratch.  bwd: dx = g[0] (sigmoid(x) - t) / rows. */
int vlp3d_bce_logits_blocks(long long n);
int vlp3d_bce_logits_fwd(const float *x, const float *t, long long rows, long long cols, double *partial, float *out,
                         void *stream);
int vlp3d_bce_logits_bwd(const float *x, const float *t, long long rows, long long cols, const float *g, float *dx,
                         void *stream);

/* Tail of get_joint_loss (lib/loss_helper/loss_joint.py:204-223) in one launch, the reference's fp32 order:
 * total[0] = ((core[0] + w_lang lang[0]) + (w_lcon lcon[0] + w_icon icon[0])) + ans[0] + cap[0]; total[1] = the contrastive sum
 * (data_dict["con_loss"]); `core` points at the core's total.  lang / (lcon and icon) / ans / cap may be NULL.
 * bwd: d (n + 4): d[0..n) = g[0] e_at (the core's n reported scalars, total at index at), d[n..] = g[0] [1, w_lang, w_lcon, w_icon]. */
int vlp3d_loss_tail_fwd(const float *core, const float *lang, const float *lcon, const float *icon, const float *ans,
                        const float *cap, float w_lang, float w_lcon, float w_icon, float *total, void *stream);
int vlp3d_loss_tail_bwd(const float *g, int n, int at, float w_lang, float w_lcon, float w_icon, float *d, void *stream);

/* Training-time scene augmentation on the device (csrc/augment.hip) — lib/joint/dataset.py:653-690 with
 * utils/utils_fn.py:28-142 (flip_augment, rotate_augment, scale_augment, translate) and
 * data/scannet/model_util_scannet.py:48-80 (rotate_aligned_boxes_along_axis); votes recomputed AFTER augmentation.
 * params: (B, vlp3d_augment_param_floats()) device floats per scene: flipx, flipy, ax, ay, az, sx, sy, sz, tx, ty, tz, 0,
 * M[9] = rotx(ax)^T roty(ay)^T rotz(az)^T row-major, 0, 0, 0 (host-drawn, reference call order).
 * augment_points: pc (B,N,C) in place (xyz; column height_col *= sz when >= 0); with inst (B,N) int32 ids in [0, I),
 *   I <= vlp3d_augment_max_instances(), also the per-instance point bounding boxes into ibox (B,I,6) int32 (ordered keys).
 * augment_votes: vote (B,N,9) = 3 copies of 0.5 (min + max) of the point's instance - x where valid[b][id], else 0;
 *   mask_f (B,N) float and / or mask_i (B,N) int64 (either may be NULL).
 * augment_boxes: boxes (B,M,6) [centre | lengths] -> out (B,M,6). */
int vlp3d_augment_param_floats(void);
int vlp3d_augment_max_instances(void);
int vlp3d_augment_points(float *pc, int B, int N, int C, int height_col, const float *params, const int *inst, int I, int *ibox,
                         void *stream);
int vlp3d_augment_votes(const float *pc, int B, int N, int C, const int *inst, int I, const int *ibox, const unsigned char *valid,
                        float *vote, float *mask_f, long long *mask_i, void *stream);
int vlp3d_augment_boxes(const float *boxes, int B, int M, const float *params, float *out, void *stream);

/* Geometry-stream glue (csrc/interpolate.hip): gather_xyz — new_xyz (B,M,3) = xyz (B,N,3)[idx (B,M)] (replaces the
 * transpose / gather_operation / transpose of pointnet2_modules.py:233-236); three_nn_weights — the inverse-distance weights
 * of pointnet2_modules.py:393-397 from three_nn's dist2 (n rows of 3): w = r / sum(r), r = 1/(sqrt(d2) + 1e-8); dist
 * (optional) = sqrt(d2).  vlp3d_sa_bn_fold_shift: vlp3d_sa_bn_fold whose running mean also tracks mean_shift (C). */
int vlp3d_gather_xyz(const float *xyz, const int *idx, int B, int N, int M, float *out, void *stream);
int vlp3d_gather_xyz_grad(const float *g, const int *idx, int B, int N, int M, float *dxyz, void *stream);
int vlp3d_three_nn_weights(const float *dist2, long long n, float *weight, float *dist, void *stream);
int vlp3d_sa_bn_fold_shift(const double *stats, int nslab, const float *gamma, const float *beta, float *running_mean,
                           float *running_var, int C, long long R, float eps, float momentum, int training, float *vec,
                           const float *mean_shift, void *stream);

/* Atomic-free backward of a grouped MLP's gather layer (bf16 configuration; pointnet2_modules.py:233-267 backward through
 * grouping_operation): vlp3d_sa_inverse builds, next to the ball query, the point -> rows map (inv_start (B*N+1), inv_rows
 * (B*M*S); cursor (B*N) scratch; crow / rowptr = the compact map or NULL); vlp3d_sa_bwd_gather_csr sums the layer-1
 * BatchNorm-backward rows of every point (Gsum (B*N, c0) scratch) and multiplies by the prepared W1^T (kpad x c0 bf16):
 * d(features) (B*N, C), every element written once. */
int vlp3d_sa_inverse(const int *idx, const void *crow, const int *rowptr, int B, int N, int M, int S, int *inv_start,
                     int *inv_rows, int *cursor, void *stream);
int vlp3d_sa_bwd_gather_csr(const void *G, const void *Y, int c0, const float *bn5, const void *WT, const void *crow,
                            const int *inv_start, const int *inv_rows, int B, int N, int C, float *Gsum, float *dfeat_pm,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VLP3D_H */
