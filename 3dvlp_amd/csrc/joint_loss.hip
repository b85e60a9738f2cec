// The training loss of the grounding step as one forward kernel + finalize and ONE backward kernel,
// instead of the reference's Python loops over (scene, sentence) with host syncs and ~250 framework launches each way:
//   vote loss          lib/loss_helper/loss_detection.py:24-72
//   objectness loss    loss_detection.py:74-113      (NEAR = FAR = 0.3, class weights 0.2 / 0.8)
//   box + sem-cls      loss_detection.py:116-258     (heading CE, heading residual Huber(1), 6-face distance Huber(0.15)
//                                                     against recover_assigned_gt_bboxes, semantic CE; positives only)
//   DIoU + reference   lib/loss_helper/loss_grounding.py:129-365 with utils/box_util.py:488-529 and
//                      lib/loss_helper/loss.py:6-17 (SoftmaxRankingLoss on the smooth / hard best-IoU labels)
//   total              lib/loss_helper/loss_joint.py:178-205:  10*(vote + 0.1*obj + box) + w_ref*ref + w_diou*diou,
//                      box = 0.1*heading_cls + heading_reg + 0.1*sem_cls + 20*distance
// Grid sections: thread per seed | thread per proposal | workgroup per (scene, sentence) row.  Every workgroup writes
// its share of the global sums to its own row of `part`; the finalize kernel adds the rows (no same-address atomics).
// Discrete decisions (nearest GT, IoU >= 0.25, arg-max) are evaluated with the arithmetic of the op-by-op form
// (3dvlp_amd/losses.py, impl="torch") in the same operation order — this file is compiled with -ffp-contract=off.
#include "common.h"

namespace {

struct JL {
  // differentiable inputs
  const float *vote_xyz, *obj_scores, *heading_scores, *heading_res, *rois, *sem_scores, *agg_xyz, *pred_center, *pred_size,
      *cluster_ref;
  // labels
  const float *seed_xyz;
  const int *seed_inds;
  const float *vote_label, *vote_mask, *center_label;
  const int *hcl;
  const float *hrl;
  const int *scl;
  const float *srl;
  const int *sem_label;
  const float *ref_center, *ref_size;
  const int *lang_num;
  const float *coin, *mean_size;
  int B, S, N, K, G, L, NH, NC;
  float near_thr, far_thr, w0, w1, w_ref, w_diou;
  int smooth;
};

enum { VOTE_NUM = 0, VOTE_DEN, OBJ_NUM, OBJ_DEN, POS, HC_NUM, HR_NUM, DIST_NUM, SEM_NUM, REF_SUM, DIOU_SUM, ACC_NUM, RATE25,
       RATE5, NSUMS = 16 };
enum { O_VOTE = 0, O_OBJ, O_HC, O_HR, O_DIST, O_SEM, O_BOX, O_REF, O_DIOU, O_TOTAL, O_POS, O_NEG, O_ACC, O_R25, O_R5, NOUT };

constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float bsum(float v, float *red) {  // 256 threads; red: 4 floats
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float bmaxf(float v, float *red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// block arg-max with first-index tie break (torch.argmax): every thread passes its own (value, smallest index)
__device__ __forceinline__ void bargmax(float &v, int &k, float *rv, int *rk) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(v, off);
    const int ok = __shfl_xor(k, off);
    if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = v; rk[threadIdx.x >> 6] = k; }
  __syncthreads();
  v = rv[0]; k = rk[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (rv[w] > v || (rv[w] == v && rk[w] < k)) { v = rv[w]; k = rk[w]; }
}

__device__ __forceinline__ float huberf(float e, float delta) {
  const float a = fabsf(e), q = fminf(a, delta);
  return 0.5f * q * q + delta * (a - q);
}

// nearest GT vote (L1) of seed (b,s)
__device__ __forceinline__ float vote_term(const JL &a, int b, int s, float &mask, float (&diff)[3]) {
  const long long bs = (long long)b * a.S + s;
  const int p = a.seed_inds[bs];
  mask = a.vote_mask[(long long)b * a.N + p];
  const float *gl = a.vote_label + ((long long)b * a.N + p) * 9;
  const float *sx = a.seed_xyz + bs * 3, *vx = a.vote_xyz + bs * 3;
  float best = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float df[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) df[c] = vx[c] - (gl[3 * j + c] + sx[c]);
    const float d = (fabsf(df[0]) + fabsf(df[1])) + fabsf(df[2]);
    if (j == 0 || d < best) {  // torch.min over the three GT votes: first minimum
      best = d;
#pragma unroll
      for (int c = 0; c < 3; ++c) diff[c] = df[c];
    }
  }
  return best;
}

// nearest GT centre (squared L2, first minimum) of proposal (b,k): nn_distance(agg, center_label)
__device__ __forceinline__ float nearest_gt(const JL &a, int b, int k, int &g) {
  const float *p = a.agg_xyz + ((long long)b * a.K + k) * 3;
  float best = 0.f;
  g = 0;
  const float px = p[0], py = p[1], pz = p[2];
  const float *cl = a.center_label + (long long)b * a.G * 3;
  // eight centres requested before the first compare (clamped addresses, no branch around a load): as a plain loop every
  // iteration waited for its own three loads — 256 dependent round trips, the longest chain of the launch; the compares keep
  // their order (first minimum)
  for (int i0 = 0; i0 < a.G; i0 += 8) {
    float cx[8], cy[8], cz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float *c = cl + (long long)min(i0 + u, a.G - 1) * 3;
      cx[u] = c[0]; cy[u] = c[1]; cz[u] = c[2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u;
      const float dx = px - cx[u], dy = py - cy[u], dz = pz - cz[u];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (i < a.G && (i == 0 || d < best)) { best = d; g = i; }
    }
  }
  return best;
}

// Box terms of a positive proposal against its assigned GT box g (recover_assigned_gt_bboxes + compute_box_loss).
struct BoxT {
  int hcl, sem;
  float res;        // heading residual error (normalised)
  float gtd[6];     // GT distances to the six faces
  float cs, sn;     // cos / sin of -gt_heading
};
__device__ __forceinline__ void box_targets(const JL &a, int b, int k, int g, BoxT &t) {
  const long long bg = (long long)b * a.G + g, bk = (long long)b * a.K + k;
  t.hcl = a.hcl[bg];
  t.sem = a.sem_label[bg];
  const float hrl = a.hrl[bg];
  t.res = a.heading_res[bk * a.NH + t.hcl] - hrl / (kPi / (float)a.NH);
  const float heading = a.NH != 1 ? (float)t.hcl * ((2.f * kPi) / (float)a.NH) + hrl : 0.f;
  t.cs = cosf(-heading);
  t.sn = sinf(-heading);
  const float *ms = a.mean_size + (long long)a.scl[bg] * 3, *sr = a.srl + bg * 3;
  const float *p = a.agg_xyz + bk * 3, *c = a.center_label + bg * 3;
  const float ox = p[0] - c[0], oy = p[1] - c[1], oz = p[2] - c[2];
  const float rot[3] = {ox * t.cs + oy * t.sn, -ox * t.sn + oy * t.cs, oz};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float half = (ms[q] + sr[q]) / 2.f;
    t.gtd[q] = half + rot[q];
    t.gtd[q + 3] = half - rot[q];
  }
}

__device__ __forceinline__ float lse_of(const float *x, int n) {
  float m = x[0];
  for (int c = 1; c < n; ++c) m = fmaxf(m, x[c]);
  float e = 0.f;
  for (int c = 0; c < n; ++c) e += expf(x[c] - m);
  return m + logf(e);
}

// box3d_diou_batch_tensor (utils/box_util.py:488-529), operation order of losses.box3d_diou_batch_tensor
__device__ __forceinline__ void diou_pair(const float *c1, const float *s1, const float *c2, const float *s2, float &iou,
                                          float &diou) {
  float e[3], o[3], dc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float lo1 = c1[q] - s1[q] / 2.f, hi1 = c1[q] + s1[q] / 2.f;
    const float lo2 = c2[q] - s2[q] / 2.f, hi2 = c2[q] + s2[q] / 2.f;
    e[q] = fmaxf(fminf(hi1, hi2) - fmaxf(lo1, lo2), 0.f);
    o[q] = fmaxf(fmaxf(hi1, hi2) - fminf(lo1, lo2), 0.f);
    dc[q] = c1[q] - c2[q];
  }
  const float area1 = s1[0] * s1[1] * s1[2], area2 = s2[0] * s2[1] * s2[2];
  const float inter = e[0] * e[1] * e[2];
  iou = inter / (area1 + area2 - inter);
  const float inter_diag = (dc[0] * dc[0] + dc[1] * dc[1]) + dc[2] * dc[2];
  const float outer_diag = (o[0] * o[0] + o[1] * o[1]) + o[2] * o[2];
  diou = fminf(fmaxf(iou - 1.5f * inter_diag / outer_diag, -1.f), 1.f);
}

// d(diou)/d(center1), d(diou)/d(size1) with torch's autograd conventions (binary max/min split ties evenly, clamp passes
// the gradient on its closed interval)
__device__ __forceinline__ void diou_grad(const float *c1, const float *s1, const float *c2, const float *s2, float (&dcen)[3],
                                          float (&dsz)[3]) {
  float e[3], o[3], dc[3], de_hi[3], de_lo[3], do_hi[3], do_lo[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float lo1 = c1[q] - s1[q] / 2.f, hi1 = c1[q] + s1[q] / 2.f;
    const float lo2 = c2[q] - s2[q] / 2.f, hi2 = c2[q] + s2[q] / 2.f;
    const float ed = fminf(hi1, hi2) - fmaxf(lo1, lo2), od = fmaxf(hi1, hi2) - fminf(lo1, lo2);
    e[q] = fmaxf(ed, 0.f);
    o[q] = fmaxf(od, 0.f);
    dc[q] = c1[q] - c2[q];
    const float pe = ed >= 0.f ? 1.f : 0.f, po = od >= 0.f ? 1.f : 0.f;
    de_hi[q] = pe * (hi1 < hi2 ? 1.f : (hi1 == hi2 ? 0.5f : 0.f));     // d e / d hi1  (hi1 is the min)
    de_lo[q] = -pe * (lo1 > lo2 ? 1.f : (lo1 == lo2 ? 0.5f : 0.f));    // d e / d lo1  (lo1 is the max)
    do_hi[q] = po * (hi1 > hi2 ? 1.f : (hi1 == hi2 ? 0.5f : 0.f));
    do_lo[q] = -po * (lo1 < lo2 ? 1.f : (lo1 == lo2 ? 0.5f : 0.f));
  }
  const float area1 = s1[0] * s1[1] * s1[2], area2 = s2[0] * s2[1] * s2[2];
  const float inter = e[0] * e[1] * e[2];
  const float uni = area1 + area2 - inter;
  const float iou = inter / uni;
  const float D = (dc[0] * dc[0] + dc[1] * dc[1]) + dc[2] * dc[2];
  const float O = (o[0] * o[0] + o[1] * o[1]) + o[2] * o[2];
  const float raw = iou - 1.5f * D / O;
  const float pass = (raw >= -1.f && raw <= 1.f) ? 1.f : 0.f;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float others = (q == 0 ? e[1] * e[2] : (q == 1 ? e[0] * e[2] : e[0] * e[1]));
    const float dA = (q == 0 ? s1[1] * s1[2] : (q == 1 ? s1[0] * s1[2] : s1[0] * s1[1]));  // d area1 / d s1[q]
    // w.r.t. centre: hi1 and lo1 both move by 1
    {
      const float dinter = others * (de_hi[q] + de_lo[q]);
      const float diou_ = (dinter * uni - inter * (-dinter)) / (uni * uni);
      const float dO = 2.f * o[q] * (do_hi[q] + do_lo[q]);
      const float dD = 2.f * dc[q];
      dcen[q] = pass * (diou_ - 1.5f * (dD * O - D * dO) / (O * O));
    }
    // w.r.t. size: hi1 moves by +1/2, lo1 by -1/2
    {
      const float dinter = others * 0.5f * (de_hi[q] - de_lo[q]);
      const float diou_ = (dinter * uni - inter * (dA - dinter)) / (uni * uni);
      const float dO = 2.f * o[q] * 0.5f * (do_hi[q] - do_lo[q]);
      dsz[q] = pass * (diou_ - 1.5f * (-D * dO) / (O * O));
    }
  }
}

// label weight t_k of proposal k in row (i,j) (loss_grounding.py:258-271) given the row's saved decisions
__device__ __forceinline__ float row_label(const JL &a, int valid, int amax, int cnt, int k, float iou_m) {
  if (!valid) return 0.f;
  if (a.smooth && cnt >= 2) return k == amax ? 0.95f : (iou_m >= 0.25f ? 0.05f / (float)(cnt - 1) : 0.f);
  return k == amax ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void jl_fwd_kernel(JL a, double *__restrict__ part, int nb_vote, int nb_prop,
                                                     int *__restrict__ assign, int *__restrict__ objlab,
                                                     int *__restrict__ rowinfo) {
  __shared__ float red[4];
  __shared__ float rv[4];
  __shared__ int rk[4];
  const int blk = blockIdx.x;
  double *sums = part + (long long)blk * NSUMS;  // this workgroup's row, fully written below
  float acc[NSUMS];
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) acc[q] = 0.f;
  if (blk < nb_vote) {  // ---- vote loss: thread per seed
    const long long t = (long long)blk * 256 + threadIdx.x;
    if (t < (long long)a.B * a.S) {
      float mask, df[3];
      const float d = vote_term(a, (int)(t / a.S), (int)(t % a.S), mask, df);
      acc[VOTE_NUM] = d * mask;
      acc[VOTE_DEN] = mask;
    }
  } else if (blk < nb_vote + nb_prop) {  // ---- objectness + box + sem-cls: thread per proposal
    const long long t = (long long)(blk - nb_vote) * 256 + threadIdx.x;
    if (t < (long long)a.B * a.K) {
      const int b = (int)(t / a.K), k = (int)(t % a.K);
      int g;
      const float euc = sqrtf(nearest_gt(a, b, k, g) + 1e-6f);
      const bool near = euc < a.near_thr, far = euc > a.far_thr;
      const float mask = (near || far) ? 1.f : 0.f;
      assign[t] = g;
      objlab[t] = (near ? 1 : 0) | ((near || far) ? 2 : 0);
      const float *sc = a.obj_scores + t * 2;
      const float lse = lse_of(sc, 2);
      acc[OBJ_NUM] = (near ? a.w1 * (lse - sc[1]) : a.w0 * (lse - sc[0])) * mask;
      acc[OBJ_DEN] = mask;
      acc[ACC_NUM] = (((sc[1] > sc[0]) == near) ? 1.f : 0.f) * mask;
      if (near) {
        BoxT bt;
        box_targets(a, b, k, g, bt);
        acc[POS] = 1.f;
        const float *hs = a.heading_scores + t * a.NH;
        acc[HC_NUM] = lse_of(hs, a.NH) - hs[bt.hcl];
        acc[HR_NUM] = huberf(bt.res, 1.f);
        const float *r = a.rois + t * 6;
        float dsum = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) dsum += huberf(r[q] - bt.gtd[q], 0.15f);
        acc[DIST_NUM] = dsum / 6.f;
        const float *ss = a.sem_scores + t * a.NC;
        acc[SEM_NUM] = lse_of(ss, a.NC) - ss[bt.sem];
      }
    }
  } else {  // ---- DIoU + reference loss: workgroup per (scene, sentence) row
    const int row = blk - nb_vote - nb_prop, b = row / a.L, j = row % a.L;
    const int ln = a.lang_num[b];
    int *info = rowinfo + (long long)row * 4;
    if (j >= ln) {  // padded sentence: no contribution (wave-uniform branch)
      if (threadIdx.x < 4) info[threadIdx.x] = 0;
    } else {
      const bool gate = a.coin[0] < 0.5f;
      const float *gc = a.ref_center + (long long)row * 3, *gs = a.ref_size + (long long)row * 3;
      const float *logit = a.cluster_ref + (long long)row * a.K;
      float bi = -1.f, bm = -1.f, mx = -3.0e38f;
      int ki = 0x7fffffff, km = 0x7fffffff;
      float cntf = 0.f;
      for (int k = threadIdx.x; k < a.K; k += 256) {
        const long long bk = (long long)b * a.K + k;
        float iou, diou;
        diou_pair(a.pred_center + bk * 3, a.pred_size + bk * 3, gc, gs, iou, diou);
        const float objm = a.obj_scores[bk * 2 + 1] > a.obj_scores[bk * 2] ? 1.f : 0.f;
        const float im = gate ? iou * objm : iou;
        if (iou > bi) { bi = iou; ki = k; }
        if (im > bm) { bm = im; km = k; }
        cntf += im >= 0.25f ? 1.f : 0.f;
        mx = fmaxf(mx, logit[k] + 1e-8f);
      }
      bargmax(bi, ki, rv, rk);
      bargmax(bm, km, rv, rk);
      if (ki >= a.K) ki = 0;  // only when every IoU is NaN (degenerate boxes): torch.argmax would also return a valid index
      if (km >= a.K) km = 0;
      const int cnt = (int)(bsum(cntf, red) + 0.5f);
      mx = bmaxf(mx, red);
      const int valid = bi >= 0.25f ? 1 : 0;
      float es = 0.f;
      for (int k = threadIdx.x; k < a.K; k += 256) es += expf((logit[k] + 1e-8f) - mx);
      es = bsum(es, red);
      float rl = 0.f, dl = 0.f;
      if (valid)
        for (int k = threadIdx.x; k < a.K; k += 256) {
          const long long bk = (long long)b * a.K + k;
          float iou, diou;
          diou_pair(a.pred_center + bk * 3, a.pred_size + bk * 3, gc, gs, iou, diou);
          const float objm = a.obj_scores[bk * 2 + 1] > a.obj_scores[bk * 2] ? 1.f : 0.f;
          const float tk = row_label(a, valid, km, cnt, k, gate ? iou * objm : iou);
          if (tk != 0.f) {
            const float p = expf((logit[k] + 1e-8f) - mx) / es;
            rl -= tk * logf(p + 1e-8f);
            dl += tk * (1.f - diou);
          }
        }
      acc[REF_SUM] = rl / (float)ln;
      acc[DIOU_SUM] = dl;
      if (threadIdx.x == 0) {
        acc[RATE25] = (float)valid;
        acc[RATE5] = bi >= 0.5f ? 1.f : 0.f;
        info[0] = valid; info[1] = ki; info[2] = km; info[3] = cnt;
      }
    }
  }
  // the sixteen sums: wave shuffles first, then ONE exchange through LDS (sixteen block reductions were 32 barriers); the same
  // order of additions as bsum
  __shared__ float redq[4][NSUMS];
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) acc[q] = wsum(acc[q]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < NSUMS; ++q) redq[threadIdx.x >> 6][q] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x < NSUMS) {
    const int q = threadIdx.x;
    sums[q] = (double)((redq[0][q] + redq[1][q]) + (redq[2][q] + redq[3][q]));
  }
}

__global__ __launch_bounds__(256) void jl_finalize_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ sums,
                                                          const int *__restrict__ lang_num, int B, int K, float w_ref,
                                                          float w_diou, float *__restrict__ out) {
  __shared__ double red[NSUMS][64];
  const int q = threadIdx.x & 15, lane = threadIdx.x >> 4;  // 16 sums x 16 row groups
  double acc = 0.0;
  for (int r = lane; r < nblocks; r += 16) acc += part[(long long)r * NSUMS + q];
  red[q][lane] = acc;
  __syncthreads();
  if (threadIdx.x < NSUMS) {
    double s = 0.0;
    for (int l = 0; l < 16; ++l) s += red[threadIdx.x][l];
    sums[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float vote = (float)sums[VOTE_NUM] / ((float)sums[VOTE_DEN] + 1e-6f);
  const float obj = (float)sums[OBJ_NUM] / ((float)sums[OBJ_DEN] + 1e-6f);
  const float den = (float)sums[POS] + 1e-6f;
  const float hc = (float)sums[HC_NUM] / den, hr = (float)sums[HR_NUM] / den, dl = (float)sums[DIST_NUM] / den,
              sem = (float)sums[SEM_NUM] / den;
  const float box = 0.1f * hc + hr + 0.1f * sem + 20.f * dl;
  const float ref = (float)sums[REF_SUM] / (float)B, diou = (float)sums[DIOU_SUM] / (float)B;
  int tot = 0;
  for (int b = 0; b < B; ++b) tot += lang_num[b];
  out[O_VOTE] = vote; out[O_OBJ] = obj; out[O_HC] = hc; out[O_HR] = hr; out[O_DIST] = dl; out[O_SEM] = sem; out[O_BOX] = box;
  out[O_REF] = ref; out[O_DIOU] = diou;
  out[O_TOTAL] = 10.f * (vote + 0.1f * obj + box) + w_ref * ref + w_diou * diou;
  out[O_POS] = (float)sums[POS] / (float)(B * K);
  out[O_NEG] = (float)sums[OBJ_DEN] / (float)(B * K) - out[O_POS];
  out[O_ACC] = (float)sums[ACC_NUM] / ((float)sums[OBJ_DEN] + 1e-6f);
  out[O_R25] = (float)sums[RATE25] / (float)(tot > 0 ? tot : 1);
  out[O_R5] = (float)sums[RATE5] / (float)(tot > 0 ? tot : 1);
}

struct JLGrad {
  float *vote, *obj, *hs, *hr, *rois, *sem, *agg, *center, *size, *ref;
};

__global__ __launch_bounds__(256) void jl_bwd_kernel(JL a, const double *__restrict__ sums, const int *__restrict__ assign,
                                                     const int *__restrict__ objlab, const int *__restrict__ rowinfo,
                                                     const float *__restrict__ gout, int nb_vote, int nb_prop, JLGrad d) {
  __shared__ float red[4];
  const float g = gout ? *gout : 1.f;
  const int blk = blockIdx.x;
  if (blk < nb_vote) {
    const long long t = (long long)blk * 256 + threadIdx.x;
    if (t >= (long long)a.B * a.S) return;
    float mask, df[3];
    vote_term(a, (int)(t / a.S), (int)(t % a.S), mask, df);
    const float c = 10.f * g * mask / ((float)sums[VOTE_DEN] + 1e-6f);
#pragma unroll
    for (int q = 0; q < 3; ++q) d.vote[t * 3 + q] = c * (df[q] > 0.f ? 1.f : (df[q] < 0.f ? -1.f : 0.f));
  } else if (blk < nb_vote + nb_prop) {
    const long long t = (long long)(blk - nb_vote) * 256 + threadIdx.x;
    if (t >= (long long)a.B * a.K) return;
    const int b = (int)(t / a.K), k = (int)(t % a.K);
    const int lab = objlab[t] & 1, msk = (objlab[t] >> 1) & 1, gi = assign[t];
    {  // objectness
      const float *sc = a.obj_scores + t * 2;
      const float m = fmaxf(sc[0], sc[1]);
      const float e0 = expf(sc[0] - m), e1 = expf(sc[1] - m);
      const float co = msk ? g * (lab ? a.w1 : a.w0) / ((float)sums[OBJ_DEN] + 1e-6f) : 0.f;  // 10 * 0.1 = 1
      d.obj[t * 2 + 0] = co * (e0 / (e0 + e1) - (lab ? 0.f : 1.f));
      d.obj[t * 2 + 1] = co * (e1 / (e0 + e1) - (lab ? 1.f : 0.f));
    }
    float dagg[3] = {0.f, 0.f, 0.f};
    if (lab) {
      BoxT bt;
      box_targets(a, b, k, gi, bt);
      const float cden = 10.f * g / ((float)sums[POS] + 1e-6f);
      const float *hs = a.heading_scores + t * a.NH;
      const float hl = lse_of(hs, a.NH);
      for (int c = 0; c < a.NH; ++c) {
        d.hs[t * a.NH + c] = cden * 0.1f * (expf(hs[c] - hl) - (c == bt.hcl ? 1.f : 0.f));
        d.hr[t * a.NH + c] = c == bt.hcl ? cden * fmaxf(-1.f, fminf(1.f, bt.res)) : 0.f;
      }
      const float *r = a.rois + t * 6;
      float dg[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        dg[q] = cden * 20.f * fmaxf(-0.15f, fminf(0.15f, r[q] - bt.gtd[q])) / 6.f;
        d.rois[t * 6 + q] = dg[q];
      }
      // gt distances depend on the vote centre: bld = half + rot, fru = half - rot (the reference keeps this path)
      const float drx = -(dg[0] - dg[3]), dry = -(dg[1] - dg[4]), drz = -(dg[2] - dg[5]);
      dagg[0] = drx * bt.cs - dry * bt.sn;
      dagg[1] = drx * bt.sn + dry * bt.cs;
      dagg[2] = drz;
      const float *ss = a.sem_scores + t * a.NC;
      const float sl = lse_of(ss, a.NC);
      for (int c = 0; c < a.NC; ++c) d.sem[t * a.NC + c] = cden * 0.1f * (expf(ss[c] - sl) - (c == bt.sem ? 1.f : 0.f));
    } else {
      for (int c = 0; c < a.NH; ++c) { d.hs[t * a.NH + c] = 0.f; d.hr[t * a.NH + c] = 0.f; }
#pragma unroll
      for (int q = 0; q < 6; ++q) d.rois[t * 6 + q] = 0.f;
      for (int c = 0; c < a.NC; ++c) d.sem[t * a.NC + c] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) d.agg[t * 3 + q] = dagg[q];
    // DIoU term: sum over the scene's sentences of t_jk * d(1 - diou_jk)
    float dcen[3] = {0.f, 0.f, 0.f}, dsz[3] = {0.f, 0.f, 0.f};
    if (a.w_diou != 0.f) {
      const bool gate = a.coin[0] < 0.5f;
      const float objm = a.obj_scores[t * 2 + 1] > a.obj_scores[t * 2] ? 1.f : 0.f;
      const float *pc = a.pred_center + t * 3, *ps = a.pred_size + t * 3;
      const int ln = a.lang_num[b];
      const float c = -g * a.w_diou / (float)a.B;
      const float pcr[3] = {pc[0], pc[1], pc[2]}, psr[3] = {ps[0], ps[1], ps[2]};
      const int nj = min(a.L, ln);
      // the sentences' rows (decisions + reference box) of four sentences requested before the first is used: as a plain loop
      // every sentence waited for its own three dependent loads; the order of the additions is unchanged
      for (int j0 = 0; j0 < nj; j0 += 4) {
        int4 inf[4];
        float gcr[4][3], gsr[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long row = (long long)b * a.L + min(j0 + u, nj - 1);
          inf[u] = *reinterpret_cast<const int4 *>(rowinfo + row * 4);
#pragma unroll
          for (int q = 0; q < 3; ++q) { gcr[u][q] = a.ref_center[row * 3 + q]; gsr[u][q] = a.ref_size[row * 3 + q]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j0 + u >= nj || !inf[u].x) continue;
          float iou, diou;
          diou_pair(pcr, psr, gcr[u], gsr[u], iou, diou);
          const float tk = row_label(a, 1, inf[u].z, inf[u].w, k, gate ? iou * objm : iou);
          if (tk == 0.f) continue;
          float gcn[3], gsz[3];
          diou_grad(pcr, psr, gcr[u], gsr[u], gcn, gsz);
#pragma unroll
          for (int q = 0; q < 3; ++q) { dcen[q] += c * tk * gcn[q]; dsz[q] += c * tk * gsz[q]; }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { d.center[t * 3 + q] = dcen[q]; d.size[t * 3 + q] = dsz[q]; }
  } else {
    const int row = blk - nb_vote - nb_prop, b = row / a.L, j = row % a.L;
    const int ln = a.lang_num[b];
    const int *info = rowinfo + (long long)row * 4;
    float *dr = d.ref + (long long)row * a.K;
    if (j >= ln || !info[0]) {  // wave-uniform
      for (int k = threadIdx.x; k < a.K; k += 256) dr[k] = 0.f;
      return;
    }
    const bool gate = a.coin[0] < 0.5f;
    const float *gc = a.ref_center + (long long)row * 3, *gs = a.ref_size + (long long)row * 3;
    const float *logit = a.cluster_ref + (long long)row * a.K;
    float mx = -3.0e38f;
    for (int k = threadIdx.x; k < a.K; k += 256) mx = fmaxf(mx, logit[k] + 1e-8f);
    mx = bmaxf(mx, red);
    float es = 0.f;
    for (int k = threadIdx.x; k < a.K; k += 256) es += expf((logit[k] + 1e-8f) - mx);
    es = bsum(es, red);
    // S = sum_k t_k p_k / (p_k + eps)
    float S = 0.f;
    for (int k = threadIdx.x; k < a.K; k += 256) {
      const long long bk = (long long)b * a.K + k;
      float iou, diou;
      diou_pair(a.pred_center + bk * 3, a.pred_size + bk * 3, gc, gs, iou, diou);
      const float objm = a.obj_scores[bk * 2 + 1] > a.obj_scores[bk * 2] ? 1.f : 0.f;
      const float tk = row_label(a, 1, info[2], info[3], k, gate ? iou * objm : iou);
      if (tk != 0.f) {
        const float p = expf((logit[k] + 1e-8f) - mx) / es;
        S += tk * p / (p + 1e-8f);
      }
    }
    S = bsum(S, red);
    const float c = g * a.w_ref / ((float)a.B * (float)ln);
    for (int k = threadIdx.x; k < a.K; k += 256) {
      const long long bk = (long long)b * a.K + k;
      float iou, diou;
      diou_pair(a.pred_center + bk * 3, a.pred_size + bk * 3, gc, gs, iou, diou);
      const float objm = a.obj_scores[bk * 2 + 1] > a.obj_scores[bk * 2] ? 1.f : 0.f;
      const float tk = row_label(a, 1, info[2], info[3], k, gate ? iou * objm : iou);
      const float p = expf((logit[k] + 1e-8f) - mx) / es;
      dr[k] = -c * (tk * p / (p + 1e-8f) - p * S);
    }
  }
}

bool bad(const JL &a) {
  return !a.vote_xyz || !a.obj_scores || !a.heading_scores || !a.heading_res || !a.rois || !a.sem_scores || !a.agg_xyz ||
         !a.pred_center || !a.pred_size || !a.cluster_ref || !a.seed_xyz || !a.seed_inds || !a.vote_label || !a.vote_mask ||
         !a.center_label || !a.hcl || !a.hrl || !a.scl || !a.srl || !a.sem_label || !a.ref_center || !a.ref_size ||
         !a.lang_num || !a.coin || !a.mean_size || a.B < 1 || a.S < 1 || a.N < 1 || a.K < 1 || a.G < 1 || a.L < 1 ||
         a.NH < 1 || a.NC < 1;
}

int nblocks_of(int B, int S, int K, int L, int &nbv, int &nbp) {
  nbv = (int)(((long long)B * S + 255) / 256);
  nbp = (int)(((long long)B * K + 255) / 256);
  return nbv + nbp + B * L;
}

}  // namespace

#define JL_ARGS                                                                                                            \
  const float *vote_xyz, const float *obj_scores, const float *heading_scores, const float *heading_res_norm,            \
      const float *rois, const float *sem_scores, const float *agg_xyz, const float *pred_center, const float *pred_size, \
      const float *cluster_ref, const float *seed_xyz, const int *seed_inds, const float *vote_label,                    \
      const float *vote_mask, const float *center_label, const int *heading_class_label,                                 \
      const float *heading_residual_label, const int *size_class_label, const float *size_residual_label,                \
      const int *sem_cls_label, const float *ref_center, const float *ref_size, const int *lang_num, const float *coin,   \
      const float *mean_size, int B, int S, int N, int K, int G, int L, int NH, int NC, float near_thr, float far_thr,    \
      float w0, float w1, float w_ref, float w_diou, int smooth_labels
#define JL_PACK                                                                                                            \
  {vote_xyz, obj_scores, heading_scores, heading_res_norm, rois, sem_scores, agg_xyz, pred_center, pred_size, cluster_ref, \
   seed_xyz, seed_inds, vote_label, vote_mask, center_label, heading_class_label, heading_residual_label,                 \
   size_class_label, size_residual_label, sem_cls_label, ref_center, ref_size, lang_num, coin, mean_size, B, S, N, K, G, L, \
   NH, NC, near_thr, far_thr, w0, w1, w_ref, w_diou, smooth_labels}

// rows of 16 doubles the caller must provide as `part` (one per workgroup of the forward kernel)
extern "C" long long vlp3d_joint_loss_rows(int B, int S, int K, int L) {
  if (B < 1 || S < 1 || K < 1 || L < 1) return 0;
  int nbv, nbp;
  return nblocks_of(B, S, K, L, nbv, nbp);
}

extern "C" int vlp3d_joint_loss_fwd(JL_ARGS, double *part, double *sums, float *out, int *assign, int *objlab, int *rowinfo,
                                    void *stream) {
  JL a = JL_PACK;
  if (bad(a) || !part || !sums || !out || !assign || !objlab || !rowinfo) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int nbv, nbp;
  const int nblocks = nblocks_of(B, S, K, L, nbv, nbp);
  hipLaunchKernelGGL(jl_fwd_kernel, dim3(nblocks), dim3(256), 0, s, a, part, nbv, nbp, assign, objlab, rowinfo);
  hipLaunchKernelGGL(jl_finalize_kernel, dim3(1), dim3(256), 0, s, part, nblocks, sums, lang_num, B, K, w_ref, w_diou, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_joint_loss_bwd(JL_ARGS, const double *sums, const int *assign, const int *objlab, const int *rowinfo,
                                    const float *gout, float *d_vote_xyz, float *d_obj_scores, float *d_heading_scores,
                                    float *d_heading_res_norm, float *d_rois, float *d_sem_scores, float *d_agg_xyz,
                                    float *d_pred_center, float *d_pred_size, float *d_cluster_ref, void *stream) {
  JL a = JL_PACK;
  if (bad(a) || !sums || !assign || !objlab || !rowinfo || !d_vote_xyz || !d_obj_scores || !d_heading_scores ||
      !d_heading_res_norm || !d_rois || !d_sem_scores || !d_agg_xyz || !d_pred_center || !d_pred_size || !d_cluster_ref)
    return VLP3D_EINVAL;
  int nbv, nbp;
  const int nblocks = nblocks_of(B, S, K, L, nbv, nbp);
  JLGrad d = {d_vote_xyz, d_obj_scores, d_heading_scores, d_heading_res_norm, d_rois, d_sem_scores, d_agg_xyz,
              d_pred_center, d_pred_size, d_cluster_ref};
  hipLaunchKernelGGL(jl_bwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, a, sums, assign, objlab, rowinfo, gout,
                     nbv, nbp, d);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// The reporting tensors the reference leaves in data_dict (loss_detection.py:101-108, loss_grounding.py:84-92) in the
// dtypes its callers expect, from the kernel's compact outputs — one launch instead of eleven framework ones:
//   object_assignment i64 (B,K) | objectness_label i64 (B,K) | objectness_mask f32 (B,K) | cluster_labels f32 (B,L,K)
namespace {
__global__ __launch_bounds__(256) void jl_report_kernel(const int *__restrict__ assign, const int *__restrict__ objlab,
                                                        const int *__restrict__ rowinfo, int B, int K, int L,
                                                        long long *__restrict__ assign64, long long *__restrict__ label64,
                                                        float *__restrict__ mask, float *__restrict__ cluster_labels) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < (long long)B * K) {
    assign64[t] = assign[t];
    label64[t] = objlab[t] & 1;
    mask[t] = (float)((objlab[t] >> 1) & 1);
  }
  if (t < (long long)B * L * K) {
    const long long row = t / K;
    const int k = (int)(t - row * K);
    cluster_labels[t] = (rowinfo[row * 4] != 0 && rowinfo[row * 4 + 1] == k) ? 1.f : 0.f;
  }
}
}  // namespace

extern "C" int vlp3d_joint_loss_report(const int *assign, const int *objlab, const int *rowinfo, int B, int K, int L,
                                       long long *assign64, long long *label64, float *mask, float *cluster_labels,
                                       void *stream) {
  if (!assign || !objlab || !rowinfo || !assign64 || !label64 || !mask || !cluster_labels || B < 1 || K < 1 || L < 1)
    return VLP3D_EINVAL;
  const long long n = (long long)B * K * (L > 1 ? L : 1);
  hipLaunchKernelGGL(jl_report_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, assign, objlab,
                     rowinfo, B, K, L, assign64, label64, mask, cluster_labels);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
