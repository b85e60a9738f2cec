// OCC / OSC InfoNCE of the contrast module for gfx950 — the core of models/constrast_module/constrast_module.py:53-131
// (NCELoss :24-37, SoftCrossEntropy :18-21, create_box_batch :9-15 + the IoU > 0.25 targets) for ALL (scene,
// sentence) pairs in three launches (forward, backward rows, backward columns) instead of ~45 framework launches
// forward and as many in autograd's backward (the reference itself: a Python double loop with a host sync per pair).
//
// Inputs are the L2-normalised projections: text (B,L,D), box (B,K,D) [pc_proj], boxi (B,K,D) [pc_proj_iou];
// obj (B,K) in {0,1} = objectness argmax; GT boxes (B,L,3)+(B,L,3) (size is grown by 1e-2 here, as the reference
// does), predicted boxes (B,K,3)+(B,K,3); lang_num (B).  With P_b = sum_k obj, T[l,k] = obj[k] * (IoU(gt_l, pred_k) >
// 0.25), log-softmax over the participating columns only:
//   OCC: sim[l,k] = text_l . box_k;    loss_v[l] = -sum_k log_softmax(sim[l,:])[k] T[l,k] / P;   out[0] = sum 0.5 loss_v / B
//   OSC: simi[k,j] = boxi_k . boxi_j;  quad[l] = sum_kj T[l,k] (-log_softmax(simi[k,:])[j]) T[l,j] / P^2;  out[1] = sum quad / B
// over sentences l < lang_num[b] of scenes with P_b > 0.
// One workgroup per row of a similarity matrix (L text rows + K proposal rows per scene): thread = column.
#include "common.h"

namespace {

constexpr int MAXC = 4;  // columns per thread: K <= 1024

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float block_sum(float v, float *red) {  // 256 threads; result to all
  v = wave_sum_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float *red) {
  v = wave_max_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// IoU of axis-aligned boxes, the expression of grounding.axis_aligned_iou / utils/box_util.py:488-529
__device__ __forceinline__ bool iou_hit(const float *__restrict__ c1, const float *__restrict__ s1raw,
                                        const float *__restrict__ c2, const float *__restrict__ s2) {
  float inter = 1.f, v1 = 1.f, v2 = 1.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float s1 = s1raw[a] + 1e-2f;
    const float lo = fmaxf(c1[a] - s1 / 2, c2[a] - s2[a] / 2);
    const float hi = fminf(c1[a] + s1 / 2, c2[a] + s2[a] / 2);
    const float e = fmaxf(hi - lo, 0.f);
    inter = a == 0 ? e : inter * e;
    v1 = a == 0 ? s1 : v1 * s1;
    v2 = a == 0 ? s2[a] : v2 * s2[a];
  }
  return inter / (v1 + v2 - inter) > 0.25f;
}

struct Args {
  const float *text, *box, *boxi, *obj, *gt_c, *gt_s, *pr_c, *pr_s;
  const long long *lang_num;
  int B, L, K, D;
};

// Shared per-row work: scores of this row against all participating columns and the row's log-sum-exp.
// row < L: text row l against box;  row >= L: proposal row k = row - L against boxi.
struct RowCtx {
  int b, row, l, k;
  bool occ;
  float P;
};

__device__ __forceinline__ void row_scores(const Args &a, const RowCtx &c, float *sq, float (&s)[MAXC], bool (&part)[MAXC]) {
  const int D = a.D;
  const float *q = c.occ ? a.text + ((long long)c.b * a.L + c.l) * D : a.boxi + ((long long)c.b * a.K + c.k) * D;
  for (int d = threadIdx.x; d < D; d += 256) sq[d] = q[d];
  __syncthreads();
  const float *keys = (c.occ ? a.box : a.boxi) + (long long)c.b * a.K * D;
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    const int j = threadIdx.x + 256 * u;
    part[u] = j < a.K && a.obj[(long long)c.b * a.K + j] != 0.f;
    float acc = 0.f;
    if (part[u]) {
      const float4 *kr = reinterpret_cast<const float4 *>(keys + (long long)j * D);
      const float4 *qq = reinterpret_cast<const float4 *>(sq);
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 kv = kr[d4], qv = qq[d4];
        acc += kv.x * qv.x;
        acc += kv.y * qv.y;
        acc += kv.z * qv.z;
        acc += kv.w * qv.w;
      }
    }
    s[u] = acc;
  }
}

// T[l][j] for this thread's columns; and (OSC rows) c_l = w_l * T[l][k] into sc[]
__device__ __forceinline__ bool target_lj(const Args &a, int b, int l, int j) {
  return a.obj[(long long)b * a.K + j] != 0.f &&
         iou_hit(a.gt_c + ((long long)b * a.L + l) * 3, a.gt_s + ((long long)b * a.L + l) * 3,
                 a.pr_c + ((long long)b * a.K + j) * 3, a.pr_s + ((long long)b * a.K + j) * 3);
}

__global__ __launch_bounds__(256) void contrast_fwd_kernel(Args a, float *__restrict__ out, float *__restrict__ lse) {
  extern __shared__ float sm[];  // [D] query, [8] reduction scratch, [L] c_l
  float *sq = sm, *red = sm + a.D, *sc = red + 8;
  RowCtx c;
  c.b = blockIdx.x / (a.L + a.K);
  c.row = blockIdx.x - c.b * (a.L + a.K);
  c.occ = c.row < a.L;
  c.l = c.row;
  c.k = c.row - a.L;
  float pc = 0.f;
  for (int j = threadIdx.x; j < a.K; j += 256) pc += a.obj[(long long)c.b * a.K + j] != 0.f ? 1.f : 0.f;
  c.P = block_sum(pc, red);
  const int nlang = (int)min((long long)a.L, a.lang_num[c.b]);
  if (c.P == 0.f || nlang <= 0 || (c.occ && c.l >= nlang)) {  // uniform: the row contributes nothing
    if (threadIdx.x == 0) lse[blockIdx.x] = 0.f;
    return;
  }
  float s[MAXC];
  bool part[MAXC];
  row_scores(a, c, sq, s, part);
  float m = -3.0e38f;
#pragma unroll
  for (int u = 0; u < MAXC; ++u)
    if (part[u]) m = fmaxf(m, s[u]);
  m = block_max(m, red);
  float e = 0.f;
#pragma unroll
  for (int u = 0; u < MAXC; ++u)
    if (part[u]) e += __expf(s[u] - m);
  const float ls = m + __logf(block_sum(e, red));
  if (threadIdx.x == 0) lse[blockIdx.x] = ls;
  if (c.occ) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u)
      if (part[u] && target_lj(a, c.b, c.l, threadIdx.x + 256 * u)) acc += ls - s[u];
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, 0.5f * tot / c.P / (float)a.B);
  } else {
    if ((int)threadIdx.x < a.L)
      sc[threadIdx.x] = ((int)threadIdx.x < nlang && target_lj(a, c.b, threadIdx.x, c.k)) ? 1.f : 0.f;
    __syncthreads();
    float any = 0.f;
    for (int l = 0; l < a.L; ++l) any += sc[l];
    if (any == 0.f) return;  // uniform: proposal k is a positive of no sentence
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u)
      if (part[u]) {
        float w = 0.f;
        for (int l = 0; l < a.L; ++l)
          if (sc[l] != 0.f && target_lj(a, c.b, l, threadIdx.x + 256 * u)) w += 1.f;
        acc += (ls - s[u]) * w;
      }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out + 1, tot / (c.P * c.P) / (float)a.B);
  }
}

// backward, rows: dS[row][j] (stored, zero where not participating) and the gradient of the row's own vector
__global__ __launch_bounds__(256) void contrast_bwd_rows_kernel(Args a, const float *__restrict__ lse,
                                                                const float *__restrict__ g_occ,
                                                                const float *__restrict__ g_osc,
                                                                float *__restrict__ dS, float *__restrict__ dtext,
                                                                float *__restrict__ dboxi) {
  extern __shared__ float sm[];  // [D] query, [8] scratch, [L] c_l, [K] dS row
  float *sq = sm, *red = sm + a.D, *sc = red + 8, *sd = sc + a.L;
  RowCtx c;
  c.b = blockIdx.x / (a.L + a.K);
  c.row = blockIdx.x - c.b * (a.L + a.K);
  c.occ = c.row < a.L;
  c.l = c.row;
  c.k = c.row - a.L;
  float *drow = dS + (long long)blockIdx.x * a.K;
  float *dq = c.occ ? dtext + ((long long)c.b * a.L + c.l) * a.D : dboxi + ((long long)c.b * a.K + c.k) * a.D;
  float pc = 0.f;
  for (int j = threadIdx.x; j < a.K; j += 256) pc += a.obj[(long long)c.b * a.K + j] != 0.f ? 1.f : 0.f;
  c.P = block_sum(pc, red);
  const int nlang = (int)min((long long)a.L, a.lang_num[c.b]);
  const float go = c.occ ? (g_occ ? *g_occ : 0.f) : (g_osc ? *g_osc : 0.f);
  bool live = c.P != 0.f && nlang > 0 && !(c.occ && c.l >= nlang) && go != 0.f;
  if (live && !c.occ) {
    if ((int)threadIdx.x < a.L)
      sc[threadIdx.x] = ((int)threadIdx.x < nlang && target_lj(a, c.b, threadIdx.x, c.k)) ? 1.f : 0.f;
    __syncthreads();
    float any = 0.f;
    for (int l = 0; l < a.L; ++l) any += sc[l];
    live = any != 0.f;
  }
  if (!live) {  // uniform
    for (int j = threadIdx.x; j < a.K; j += 256) drow[j] = 0.f;
    for (int d = threadIdx.x; d < a.D; d += 256) dq[d] = 0.f;
    return;
  }
  float s[MAXC], w[MAXC];
  bool part[MAXC];
  row_scores(a, c, sq, s, part);
  const float ls = lse[blockIdx.x];
  float wsum = 0.f;
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    w[u] = 0.f;
    if (part[u]) {
      const int j = threadIdx.x + 256 * u;
      if (c.occ) {
        w[u] = target_lj(a, c.b, c.l, j) ? 1.f : 0.f;
      } else {
        for (int l = 0; l < a.L; ++l)
          if (sc[l] != 0.f && target_lj(a, c.b, l, j)) w[u] += 1.f;
      }
      wsum += w[u];
    }
  }
  wsum = block_sum(wsum, red);
  // OCC: loss = 0.5/(B P) sum_j w_j (lse - s_j);  OSC: loss = 1/(B P^2) sum_j w_j (lse - s_j)
  const float coef = go * (c.occ ? 0.5f / (c.P * (float)a.B) : 1.0f / (c.P * c.P * (float)a.B));
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    const int j = threadIdx.x + 256 * u;
    if (j < a.K) {
      const float v = part[u] ? coef * (__expf(s[u] - ls) * wsum - w[u]) : 0.f;
      drow[j] = v;
      sd[j] = v;
    }
  }
  __syncthreads();
  const float *keys = (c.occ ? a.box : a.boxi) + (long long)c.b * a.K * a.D;
  for (int d = threadIdx.x; d < a.D; d += 256) {
    float acc = 0.f;
    int j = 0;
    for (; j + 8 <= a.K; j += 8) {  // eight key rows in flight (one row per iteration is a chain of 256 load latencies)
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = keys[(long long)(j + u) * a.D + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += sd[j + u] * x[u];
    }
    for (; j < a.K; ++j) acc += sd[j] * keys[(long long)j * a.D + d];
    dq[d] = acc;
  }
}

// backward, columns: dbox[b,j] = sum_l dS[l][j] text_l;   dboxi[b,j] += sum_k dS[L+k][j] boxi_k
__global__ __launch_bounds__(128) void contrast_bwd_cols_kernel(Args a, const float *__restrict__ dS,
                                                                float *__restrict__ dbox, float *__restrict__ dboxi) {
  const int b = blockIdx.x / a.K, j = blockIdx.x - b * a.K;
  const float *ds = dS + (long long)b * (a.L + a.K) * a.K + j;
  for (int d = threadIdx.x; d < a.D; d += 128) {
    float acc = 0.f;
    for (int l = 0; l < a.L; ++l) acc += ds[(long long)l * a.K] * a.text[((long long)b * a.L + l) * a.D + d];
    dbox[((long long)b * a.K + j) * a.D + d] = acc;
    float acc2 = 0.f;
    int k = 0;
    for (; k + 8 <= a.K; k += 8) {  // eight rows in flight (one row per iteration is a chain of load latencies)
      float v[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = ds[(long long)(a.L + k + u) * a.K];
        x[u] = a.boxi[((long long)b * a.K + k + u) * a.D + d];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc2 += v[u] * x[u];
    }
    for (; k < a.K; ++k) acc2 += ds[(long long)(a.L + k) * a.K] * a.boxi[((long long)b * a.K + k) * a.D + d];
    dboxi[((long long)b * a.K + j) * a.D + d] += acc2;
  }
}

bool bad_args(const Args &a) {
  return !a.text || !a.box || !a.boxi || !a.obj || !a.gt_c || !a.gt_s || !a.pr_c || !a.pr_s || !a.lang_num || a.B < 1 ||
         a.L < 1 || a.L > 64 || a.K < 1 || a.K > 256 * MAXC || a.D < 4 || (a.D & 3) || a.D > 1024;
}

}  // namespace

extern "C" int vlp3d_contrast_fwd(const float *text, const float *box, const float *boxi, const float *obj,
                                  const float *gt_center, const float *gt_size, const float *pred_center,
                                  const float *pred_size, const long long *lang_num, int B, int L, int K, int D,
                                  float *out2, float *lse, void *stream) {
  Args a = {text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num, B, L, K, D};
  if (bad_args(a) || !out2 || !lse) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = vlp3d_zero_words(out2, 2, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(contrast_fwd_kernel, dim3(B * (L + K)), dim3(256), sizeof(float) * (D + 8 + L), s, a, out2, lse);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_contrast_bwd(const float *text, const float *box, const float *boxi, const float *obj,
                                  const float *gt_center, const float *gt_size, const float *pred_center,
                                  const float *pred_size, const long long *lang_num, int B, int L, int K, int D,
                                  const float *lse, const float *g_occ, const float *g_osc, float *dS, float *dtext,
                                  float *dbox, float *dboxi, void *stream) {
  Args a = {text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num, B, L, K, D};
  if (bad_args(a) || !lse || !dS || !dtext || !dbox || !dboxi) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(contrast_bwd_rows_kernel, dim3(B * (L + K)), dim3(256), sizeof(float) * (D + 8 + L + K), s, a, lse,
                     g_occ, g_osc, dS, dtext, dboxi);
  hipLaunchKernelGGL(contrast_bwd_cols_kernel, dim3(B * K), dim3(128), 0, s, a, dS, dbox, dboxi);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
