// The reduced grounding loss of grounding_step.grounding_loss as one forward kernel (+ finalize) and one backward
// kernel, instead of ~60 framework launches forward and as many in autograd's backward.  Components:
//   vote loss        lib/loss_helper/loss_detection.py:24-72   sum_seeds min_j |vote - gt_vote_j|_1 * mask / (sum mask + 1e-6)
//   objectness loss  loss_detection.py:74-110  weighted CE (0.2, 0.8) of the near (<0.3) / far (>0.6) proposals,
//                    label and assignment from the nearest GT centre (nn_distance, squared L2, first minimum)
//   centre loss      Huber(0.15) of pred_center - assigned GT centre over the near proposals
//   reference loss   CE of cluster_ref rows against the proposal nearest to the referred GT centre (mean over rows)
//   total = vote + w_obj*obj + centre + w_ref*ref
// Sections of the grid: thread per seed | thread per proposal | workgroup per (scene, sentence) row.
// Every workgroup writes its share of the seven global sums to its own row of a partial buffer; the finalize kernel
// adds the rows (a few hundred workgroups adding atomically into seven addresses serialise at L2).  `sums` is kept
// for backward.
#include "common.h"

namespace {

struct LossArgs {
  // vote
  const float *vote_xyz, *seed_xyz, *vote_label, *vote_mask;  // (B,S,3) (B,S,3) (B,N,9) (B,N) f32
  const int *seed_inds;                                      // (B,S) i32 into N
  // objectness / centre
  const float *agg_xyz, *center_label, *obj_scores, *pred_center;  // (B,K,3) (B,G,3) (B,K,2) (B,K,3)
  // reference
  const float *cluster_ref, *ref_center;  // (B*L,K) (B,L,3)
  int B, S, N, K, G, L;
  float near_thr, far_thr, w0, w1, huber_delta, w_obj, w_ref;
};

enum { VOTE_NUM = 0, VOTE_DEN, OBJ_NUM, OBJ_DEN, CTR_NUM, CTR_DEN, REF_SUM, NSUMS };

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float bsum(float v, float *red) {  // 256 threads
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// nearest GT vote (L1) of seed (b,s): returns the distance, the winning j and the mask
__device__ __forceinline__ float vote_term(const LossArgs &a, int b, int s, int &jbest, float &mask, float (&diff)[3]) {
  const long long bs = (long long)b * a.S + s;
  const int p = a.seed_inds[bs];
  mask = a.vote_mask[(long long)b * a.N + p];
  const float *gl = a.vote_label + ((long long)b * a.N + p) * 9;
  const float *sx = a.seed_xyz + bs * 3, *vx = a.vote_xyz + bs * 3;
  float best = 0.f;
  jbest = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float d = 0.f, df[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      df[c] = vx[c] - (gl[3 * j + c] + sx[c]);
      d += fabsf(df[c]);
    }
    if (j == 0 || d < best) {  // torch.min over the three GT votes: first minimum
      best = d;
      jbest = j;
#pragma unroll
      for (int c = 0; c < 3; ++c) diff[c] = df[c];
    }
  }
  return best;
}

// nearest GT centre (squared L2, first minimum) of proposal (b,k)
__device__ __forceinline__ float nearest_gt(const LossArgs &a, int b, int k, int &g) {
  const float *p = a.agg_xyz + ((long long)b * a.K + k) * 3;
  float best = 0.f;
  g = 0;
  for (int i = 0; i < a.G; ++i) {
    const float *c = a.center_label + ((long long)b * a.G + i) * 3;
    const float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
    const float d = dx * dx + dy * dy + dz * dz;
    if (i == 0 || d < best) {
      best = d;
      g = i;
    }
  }
  return best;
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(LossArgs a, double *__restrict__ part, int nb_vote, int nb_prop) {
  __shared__ float red[8];
  const int blk = blockIdx.x;
  double *sums = part + (long long)blk * NSUMS;  // this workgroup's row, fully written below
  if (blk < nb_vote) {  // ---- vote loss: thread per seed
    const long long t = (long long)blk * 256 + threadIdx.x;
    float num = 0.f, den = 0.f;
    if (t < (long long)a.B * a.S) {
      int j;
      float mask, df[3];
      const float d = vote_term(a, (int)(t / a.S), (int)(t % a.S), j, mask, df);
      num = d * mask;
      den = mask;
    }
    num = bsum(num, red);
    den = bsum(den, red);
    if (threadIdx.x == 0) {
      for (int q = 0; q < NSUMS; ++q) sums[q] = 0.0;  // the whole row: no memset of the partial buffer needed
      sums[VOTE_NUM] = (double)num;
      sums[VOTE_DEN] = (double)den;
    }
  } else if (blk < nb_vote + nb_prop) {  // ---- objectness + centre: thread per proposal
    const long long t = (long long)(blk - nb_vote) * 256 + threadIdx.x;
    float on = 0.f, od = 0.f, cn = 0.f, cd = 0.f;
    if (t < (long long)a.B * a.K) {
      const int b = (int)(t / a.K), k = (int)(t % a.K);
      int g;
      const float euc = sqrtf(nearest_gt(a, b, k, g) + 1e-6f);
      const bool near = euc < a.near_thr, far = euc > a.far_thr;
      const float *sc = a.obj_scores + t * 2;
      const float m = fmaxf(sc[0], sc[1]);
      const float lse = m + logf(expf(sc[0] - m) + expf(sc[1] - m));
      const float ce = near ? a.w1 * (lse - sc[1]) : a.w0 * (lse - sc[0]);
      const float mask = (near || far) ? 1.f : 0.f;
      on = ce * mask;
      od = mask;
      if (near) {
        const float *pc = a.pred_center + t * 3, *gc = a.center_label + ((long long)b * a.G + g) * 3;
        float h = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float ae = fabsf(pc[c] - gc[c]);
          const float q = fminf(ae, a.huber_delta);
          h += 0.5f * q * q + a.huber_delta * (ae - q);
        }
        cn = h;
        cd = 1.f;
      }
    }
    on = bsum(on, red); od = bsum(od, red); cn = bsum(cn, red); cd = bsum(cd, red);
    if (threadIdx.x == 0) {
      for (int q = 0; q < NSUMS; ++q) sums[q] = 0.0;
      sums[OBJ_NUM] = (double)on;
      sums[OBJ_DEN] = (double)od;
      sums[CTR_NUM] = (double)cn;
      sums[CTR_DEN] = (double)cd;
    }
  } else {  // ---- reference loss: workgroup per (scene, sentence)
    const int row = blk - nb_vote - nb_prop, b = row / a.L;
    const float *rc = a.ref_center + (long long)row * 3;
    const float *logit = a.cluster_ref + (long long)row * a.K;
    // target = argmin_k |pred_center_k - ref|^2 (first minimum); log-sum-exp of the row
    float bd = 3.0e38f, m = -3.0e38f;
    int bk = 0x7fffffff;
    for (int k = threadIdx.x; k < a.K; k += 256) {
      const float *pc = a.pred_center + ((long long)b * a.K + k) * 3;
      const float dx = pc[0] - rc[0], dy = pc[1] - rc[1], dz = pc[2] - rc[2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < bd) { bd = d; bk = k; }
      m = fmaxf(m, logit[k]);
    }
    // block argmin with first-index tie break, block max
    __shared__ float sd[256];
    __shared__ int sk[256];
    __shared__ float smax[256];
    sd[threadIdx.x] = bd; sk[threadIdx.x] = bk; smax[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) {
        const float od = sd[threadIdx.x + off];
        const int ok = sk[threadIdx.x + off];
        if (od < sd[threadIdx.x] || (od == sd[threadIdx.x] && ok < sk[threadIdx.x])) { sd[threadIdx.x] = od; sk[threadIdx.x] = ok; }
        smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
      }
      __syncthreads();
    }
    const int target = sk[0];
    const float mx = smax[0];
    float e = 0.f;
    for (int k = threadIdx.x; k < a.K; k += 256) e += expf(logit[k] - mx);
    e = bsum(e, red);
    if (threadIdx.x == 0) {
      for (int q = 0; q < NSUMS; ++q) sums[q] = 0.0;
      sums[REF_SUM] = (double)(mx + logf(e) - logit[target]);
    }
  }
}

// sums[q] = sum over the workgroup rows of part[.][q];  out = [vote, objectness, centre, reference, total]
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ sums,
                                                            float w_obj, float w_ref, int rows, float *__restrict__ out) {
  __shared__ double red[NSUMS][256];
  double acc[NSUMS];
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) acc[q] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
    for (int q = 0; q < NSUMS; ++q) acc[q] += part[(long long)b * NSUMS + q];
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) red[q][threadIdx.x] = acc[q];
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int q = 0; q < NSUMS; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) sums[q] = red[q][0];
  const float vote = (float)sums[VOTE_NUM] / ((float)sums[VOTE_DEN] + 1e-6f);
  const float obj = (float)sums[OBJ_NUM] / ((float)sums[OBJ_DEN] + 1e-6f);
  const float ctr = (float)sums[CTR_NUM] / ((float)sums[CTR_DEN] + 1e-6f);
  const float ref = (float)sums[REF_SUM] / (float)rows;
  out[0] = vote; out[1] = obj; out[2] = ctr; out[3] = ref;
  out[4] = vote + w_obj * obj + ctr + w_ref * ref;
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(LossArgs a, const double *__restrict__ sums,
                                                       const float *__restrict__ gout, int nb_vote, int nb_prop,
                                                       float *__restrict__ d_vote, float *__restrict__ d_obj,
                                                       float *__restrict__ d_center, float *__restrict__ d_ref) {
  __shared__ float red[8];
  const float g = gout ? *gout : 1.f;
  const int blk = blockIdx.x;
  if (blk < nb_vote) {
    const long long t = (long long)blk * 256 + threadIdx.x;
    if (t >= (long long)a.B * a.S) return;
    int j;
    float mask, df[3];
    vote_term(a, (int)(t / a.S), (int)(t % a.S), j, mask, df);
    const float c = g * mask / ((float)sums[VOTE_DEN] + 1e-6f);
#pragma unroll
    for (int q = 0; q < 3; ++q) d_vote[t * 3 + q] = c * (df[q] > 0.f ? 1.f : (df[q] < 0.f ? -1.f : 0.f));
  } else if (blk < nb_vote + nb_prop) {
    const long long t = (long long)(blk - nb_vote) * 256 + threadIdx.x;
    if (t >= (long long)a.B * a.K) return;
    const int b = (int)(t / a.K), k = (int)(t % a.K);
    int gi;
    const float euc = sqrtf(nearest_gt(a, b, k, gi) + 1e-6f);
    const bool near = euc < a.near_thr, far = euc > a.far_thr;
    const float *sc = a.obj_scores + t * 2;
    const float m = fmaxf(sc[0], sc[1]);
    const float e0 = expf(sc[0] - m), e1 = expf(sc[1] - m);
    const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    const float co = (near || far) ? g * a.w_obj * (near ? a.w1 : a.w0) / ((float)sums[OBJ_DEN] + 1e-6f) : 0.f;
    d_obj[t * 2 + 0] = co * (p0 - (near ? 0.f : 1.f));
    d_obj[t * 2 + 1] = co * (p1 - (near ? 1.f : 0.f));
    // centre (directly) — the reference-loss target uses pred_center detached, so no other contribution
    const float cc = near ? g / ((float)sums[CTR_DEN] + 1e-6f) : 0.f;
    const float *pc = a.pred_center + t * 3, *gc = a.center_label + ((long long)b * a.G + gi) * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float e = pc[q] - gc[q];
      d_center[t * 3 + q] = cc * fmaxf(-a.huber_delta, fminf(a.huber_delta, e));
    }
  } else {
    const int row = blk - nb_vote - nb_prop, b = row / a.L;
    const float *rc = a.ref_center + (long long)row * 3;
    const float *logit = a.cluster_ref + (long long)row * a.K;
    float bd = 3.0e38f, m = -3.0e38f;
    int bk = 0x7fffffff;
    for (int k = threadIdx.x; k < a.K; k += 256) {
      const float *pc = a.pred_center + ((long long)b * a.K + k) * 3;
      const float dx = pc[0] - rc[0], dy = pc[1] - rc[1], dz = pc[2] - rc[2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < bd) { bd = d; bk = k; }
      m = fmaxf(m, logit[k]);
    }
    __shared__ float sd[256];
    __shared__ int sk[256];
    __shared__ float smax[256];
    sd[threadIdx.x] = bd; sk[threadIdx.x] = bk; smax[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) {
        const float od = sd[threadIdx.x + off];
        const int ok = sk[threadIdx.x + off];
        if (od < sd[threadIdx.x] || (od == sd[threadIdx.x] && ok < sk[threadIdx.x])) { sd[threadIdx.x] = od; sk[threadIdx.x] = ok; }
        smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
      }
      __syncthreads();
    }
    const int target = sk[0];
    const float mx = smax[0];
    float e = 0.f;
    for (int k = threadIdx.x; k < a.K; k += 256) e += expf(logit[k] - mx);
    e = bsum(e, red);
    const float c = g * a.w_ref / (float)(a.B * a.L);
    for (int k = threadIdx.x; k < a.K; k += 256)
      d_ref[(long long)row * a.K + k] = c * (expf(logit[k] - mx) / e - (k == target ? 1.f : 0.f));
  }
}

bool bad(const LossArgs &a) {
  return !a.vote_xyz || !a.seed_xyz || !a.vote_label || !a.vote_mask || !a.seed_inds || !a.agg_xyz || !a.center_label ||
         !a.obj_scores || !a.pred_center || !a.cluster_ref || !a.ref_center || a.B < 1 || a.S < 1 || a.N < 1 || a.K < 1 ||
         a.G < 1 || a.L < 1;
}

}  // namespace

// doubles the caller must provide as `sums`: 7 totals + 7 per workgroup of the forward kernel
extern "C" long long vlp3d_grounding_loss_sums(int B, int S, int K, int L) {
  if (B < 1 || S < 1 || K < 1 || L < 1) return 0;
  const long long nblocks = ((long long)B * S + 255) / 256 + ((long long)B * K + 255) / 256 + (long long)B * L;
  return NSUMS * (1 + nblocks);
}

extern "C" int vlp3d_grounding_loss_fwd(const float *vote_xyz, const float *seed_xyz, const int *seed_inds,
                                        const float *vote_label, const float *vote_mask, const float *agg_xyz,
                                        const float *center_label, const float *obj_scores, const float *pred_center,
                                        const float *cluster_ref, const float *ref_center, int B, int S, int N, int K,
                                        int G, int L, float near_thr, float far_thr, float w0, float w1,
                                        float huber_delta, float w_obj, float w_ref, double *sums, float *out5,
                                        void *stream) {
  LossArgs a = {vote_xyz, seed_xyz, vote_label, vote_mask, seed_inds, agg_xyz, center_label, obj_scores, pred_center,
                cluster_ref, ref_center, B, S, N, K, G, L, near_thr, far_thr, w0, w1, huber_delta, w_obj, w_ref};
  if (bad(a) || !sums || !out5) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nbv = (int)(((long long)B * S + 255) / 256), nbp = (int)(((long long)B * K + 255) / 256);
  const int nblocks = nbv + nbp + B * L;
  double *part = sums + NSUMS;  // sums: [7 totals | nblocks x 7 workgroup rows]
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(nblocks), dim3(256), 0, s, a, part, nbv, nbp);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, part, nblocks, sums, w_obj, w_ref, B * L, out5);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_grounding_loss_bwd(const float *vote_xyz, const float *seed_xyz, const int *seed_inds,
                                        const float *vote_label, const float *vote_mask, const float *agg_xyz,
                                        const float *center_label, const float *obj_scores, const float *pred_center,
                                        const float *cluster_ref, const float *ref_center, int B, int S, int N, int K,
                                        int G, int L, float near_thr, float far_thr, float w0, float w1,
                                        float huber_delta, float w_obj, float w_ref, const double *sums,
                                        const float *gout, float *d_vote, float *d_obj, float *d_center, float *d_ref,
                                        void *stream) {
  LossArgs a = {vote_xyz, seed_xyz, vote_label, vote_mask, seed_inds, agg_xyz, center_label, obj_scores, pred_center,
                cluster_ref, ref_center, B, S, N, K, G, L, near_thr, far_thr, w0, w1, huber_delta, w_obj, w_ref};
  if (bad(a) || !sums || !d_vote || !d_obj || !d_center || !d_ref) return VLP3D_EINVAL;
  const int nbv = (int)(((long long)B * S + 255) / 256), nbp = (int)(((long long)B * K + 255) / 256);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(nbv + nbp + B * L), dim3(256), 0, (hipStream_t)stream, a, sums, gout, nbv, nbp,
                     d_vote, d_obj, d_center, d_ref);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
