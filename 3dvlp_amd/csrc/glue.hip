// Small fused "glue" kernels: each replaces a dozen element-wise / index framework launches (~5 us apiece inside the
// replayed graph) between the matrix-core kernels of the grounding step.
//   roi_split        roi_heads.py:135-147   column blocks of the merged predictor output -> contiguous tensors (+ exp, scale,
//                                           arg-max masks); backward = one gather of the five gradient pieces
//   vote_epilogue    voting_module.py:51-58 + jointnet.py:148-149   vote_xyz = seed_xyz + offset, vote_features =
//                                           (seed_features + residual) / |.|_2   (vote_factor 1), and its backward
//   l2norm_rows      F.normalize(x, dim=-1) of constrast_module.py:97-117 (eps 1e-12), and its backward
//   relation_inputs  relation_module.py:95-122   object "multiview" feature rows (the reference's indexing quirk), box
//                                           centre + corner offsets (27), corner mean (no gradient: detached inputs)
//   copy_paste       match_module.py:97-121  object-feature copy-paste augmentation as a fixed-shape gather, and its adjoint
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- roi_split -------------------------------------------------------------------------------------------------------
// out (R x ld): [heading_reg NH | heading_cls NH | box 6 | objectness 2 | sem NC]
__global__ __launch_bounds__(256) void roi_split_kernel(const float *__restrict__ out, int ld, long long R, int NH, int NC,
                                                        float res_scale, float *__restrict__ hreg, float *__restrict__ hres,
                                                        float *__restrict__ hcls, float *__restrict__ rois,
                                                        float *__restrict__ obj, float *__restrict__ sem,
                                                        long long *__restrict__ obj_mask, long long *__restrict__ sem_arg) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float *o = out + r * ld;
  for (int c = 0; c < NH; ++c) {
    hreg[r * NH + c] = o[c];
    hres[r * NH + c] = o[c] * res_scale;
    hcls[r * NH + c] = o[NH + c];
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) rois[r * 6 + c] = expf(o[2 * NH + c]);
  const float o0 = o[2 * NH + 6], o1 = o[2 * NH + 7];
  obj[r * 2] = o0;
  obj[r * 2 + 1] = o1;
  obj_mask[r] = o1 > o0 ? 1 : 0;  // argmax, first maximum
  int best = 0;
  float bv = o[2 * NH + 8];
  for (int c = 0; c < NC; ++c) {
    const float v = o[2 * NH + 8 + c];
    sem[r * NC + c] = v;
    if (v > bv) { bv = v; best = c; }
  }
  sem_arg[r] = best;
}

// d(out) from the gradients of the pieces (any may be NULL = zero); columns past the 2NH + 8 + NC used ones are zeroed
__global__ __launch_bounds__(256) void roi_split_bwd_kernel(const float *__restrict__ d_hreg, const float *__restrict__ d_hres,
                                                            const float *__restrict__ d_hcls, const float *__restrict__ d_rois,
                                                            const float *__restrict__ d_obj, const float *__restrict__ d_sem,
                                                            const float *__restrict__ rois, long long R, int NH, int NC,
                                                            float res_scale, float *__restrict__ d_out, int ld) {
  // a thread per ELEMENT of d_out (a thread per row walked ~50 dependent, uncoalesced loads on 8 workgroups: 13.8 us in-step)
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= R * ld) return;
  const long long r = e / ld;
  const int c = (int)(e - r * ld);
  float v = 0.f;
  if (c < NH) {
    v = (d_hreg ? d_hreg[r * NH + c] : 0.f) + (d_hres ? d_hres[r * NH + c] * res_scale : 0.f);
  } else if (c < 2 * NH) {
    v = d_hcls ? d_hcls[r * NH + (c - NH)] : 0.f;
  } else if (c < 2 * NH + 6) {
    const int q = c - 2 * NH;
    v = d_rois ? d_rois[r * 6 + q] * rois[r * 6 + q] : 0.f;
  } else if (c < 2 * NH + 8) {
    v = d_obj ? d_obj[r * 2 + (c - 2 * NH - 6)] : 0.f;
  } else if (c < 2 * NH + 8 + NC) {
    v = d_sem ? d_sem[r * NC + (c - 2 * NH - 8)] : 0.f;
  }
  d_out[e] = v;
}

// ---- vote_epilogue: one wave per seed ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vote_epilogue_kernel(const float *__restrict__ seed_xyz, const float *__restrict__ seed_f,
                                                            const float *__restrict__ net, int ld, long long R, int C,
                                                            float *__restrict__ vote_xyz, float *__restrict__ vote_f,
                                                            float *__restrict__ norm) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float *n = net + r * ld;
  if (lane < 3) vote_xyz[r * 3 + lane] = seed_xyz[r * 3 + lane] + n[lane];
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = seed_f[r * C + c] + n[3 + c];
    ss += v * v;
  }
  const float nr = sqrtf(wave_sum(ss));
  for (int c = lane; c < C; c += 64) vote_f[r * C + c] = (seed_f[r * C + c] + n[3 + c]) / nr;
  if (lane == 0) norm[r] = nr;
}

// v = out * norm;  d_v = (g - out * <out, g>) / norm;  d_seed_f = d_v (+ nothing else here), d_net = [d_vote_xyz | d_v | 0]
__global__ __launch_bounds__(256) void vote_epilogue_bwd_kernel(const float *__restrict__ d_xyz, const float *__restrict__ d_f,
                                                                const float *__restrict__ vote_f, const float *__restrict__ norm,
                                                                long long R, int C, float *__restrict__ d_seed_f,
                                                                float *__restrict__ d_net, int ld) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float dot = 0.f;
  if (d_f)
    for (int c = lane; c < C; c += 64) dot += vote_f[r * C + c] * d_f[r * C + c];
  dot = wave_sum(dot);
  const float inv = 1.f / norm[r];
  float *dn = d_net + r * ld;
  if (lane < 3) dn[lane] = d_xyz ? d_xyz[r * 3 + lane] : 0.f;
  for (int c = lane; c < C; c += 64) {
    const float dv = d_f ? (d_f[r * C + c] - vote_f[r * C + c] * dot) * inv : 0.f;
    d_seed_f[r * C + c] = dv;
    dn[3 + c] = dv;
  }
  for (int c = 3 + C + lane; c < ld; c += 64) dn[c] = 0.f;
}

// ---- l2norm_rows: one wave per row ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float *__restrict__ x, long long R, int C, float eps,
                                                          float *__restrict__ y, float *__restrict__ norm) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += x[r * C + c] * x[r * C + c];
  const float nr = fmaxf(sqrtf(wave_sum(ss)), eps);  // F.normalize: x / max(|x|, eps)
  for (int c = lane; c < C; c += 64) y[r * C + c] = x[r * C + c] / nr;
  if (lane == 0) norm[r] = nr;
}
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                              const float *__restrict__ norm, long long R, int C, float eps,
                                                              float *__restrict__ dx) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += y[r * C + c] * g[r * C + c];
  dot = wave_sum(dot);
  const float nr = norm[r];
  const float k = nr > eps ? dot : 0.f;  // clamped norm: the denominator is a constant, no projection term
  for (int c = lane; c < C; c += 64) dx[r * C + c] = (g[r * C + c] - y[r * C + c] * k) / nr;
}

// ---- relation_inputs: one wave per proposal ---------------------------------------------------------------------------
// obj_feat[b,k,:]: 128 consecutive elements of the channel-major (B,128,N) multiview block starting at flat element
// (src + 128 b) * 128, src = seed_inds[b][vote_inds[b][k]] (relation_module.py:98-113: ids offset by b*128, rows taken
// from the channel-major reshape) read from the point-major point cloud pc (B,N,3+3+128..): element e of that block ->
// batch e / (128 N), channel (e % (128 N)) / N, point e % N, stored at pc[batch][point][6 + channel].
// (PB: pc holds bf16 rows — the loader's bf16 copy of the cloud's feature channels — widened on the way out)
template <bool PB>
__global__ __launch_bounds__(256) void relation_inputs_kernel(const void *__restrict__ pc_, int Cpc, int col0, int N,
                                                              const int *__restrict__ seed_inds, int S,
                                                              const int *__restrict__ vote_inds, const float *__restrict__ corners,
                                                              int B, int K, float *__restrict__ obj_feat,
                                                              float *__restrict__ bbox_feat, float *__restrict__ centre) {
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= (long long)B * K) return;
  const int b = (int)(t / K);
  const int src = seed_inds[(long long)b * S + vote_inds[t]];
  const long long row_id = (long long)src + (long long)b * 128;
  for (int c = lane; c < 128; c += 64) {
    const long long flat = row_id * 128 + c;
    const long long fb = flat / (128ll * N), rem = flat - fb * 128ll * N;
    const long long ch = rem / N, pt = rem - ch * N;
    float v = 0.f;
    if (fb < B) {
      const long long o = (fb * N + pt) * Cpc + col0 + ch;
      if (PB) v = __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(pc_)[o] << 16);
      else v = reinterpret_cast<const float *>(pc_)[o];
    }
    obj_feat[t * 128 + c] = v;
  }
  if (lane < 3) {  // per coordinate: min / max / mean over the 8 corners
    const float *cr = corners + t * 24 + lane;
    float mn = cr[0], mx = cr[0], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = cr[3 * j];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
      sum += v;
    }
    const float bc = (mn + mx) / 2.f;
    bbox_feat[t * 27 + lane] = bc;
#pragma unroll
    for (int j = 0; j < 8; ++j) bbox_feat[t * 27 + 3 + 3 * j + lane] = cr[3 * j] - bc;
    centre[t * 3 + lane] = sum / 8.f;
  }
}

// ---- copy_paste -------------------------------------------------------------------------------------------------------
// src[b*K + k] = flat source slot of proposal (b,k) after the augmentation (its own slot when nothing is pasted):
// background slot of rank r in scene i  <-  pool[(J_i + r) mod total]  if r < total - n_i  (match_module.py:97-121),
// J_i = objects in scenes 0..i, pool = object proposals in (scene, proposal) order.  One workgroup, B*K <= 8192.
__global__ __launch_bounds__(1024) void copy_paste_map_kernel(const long long *__restrict__ obj_mask, int B, int K,
                                                              const float *__restrict__ coin, int *__restrict__ src) {
  // One WAVE per scene (scenes beyond the 16 waves: a second round); a lane owns K/64 consecutive slots, prefix counts of
  // the object flags by a shuffle scan (the first version walked the K slots of a scene in one thread, three times: 34 us).
  extern __shared__ int sm[];  // [B*K] pool positions; [B] n_obj; [B*K] exclusive object prefix inside the scene
  int *pool = sm, *nobj = sm + B * K, *pre = nobj + B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int per = (K + 63) / 64;
  for (int b = wave; b < B; b += nwaves) {
    int c = 0;
    for (int j = 0; j < per; ++j) {
      const int k = lane * per + j;
      if (k < K) {
        pre[b * K + k] = c;  // objects before slot k among this lane's slots (lane offset added below)
        c += obj_mask[b * K + k] != 0;
      }
    }
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    const int before = inc - c;
    for (int j = 0; j < per; ++j) {
      const int k = lane * per + j;
      if (k < K) pre[b * K + k] += before;
    }
    if (lane == 63) nobj[b] = inc;
  }
  __syncthreads();
  const int n = B * K;
  int total = 0;
  for (int b = 0; b < B; ++b) total += nobj[b];
  const bool use = coin[0] < 0.5f;
  // pool: object slots in (scene, proposal) order
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int b = i / K;
    if (obj_mask[i] != 0) {
      int jb = 0;
      for (int q = 0; q < b; ++q) jb += nobj[q];  // objects in the scenes before b
      pool[jb + pre[i]] = i;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int b = i / K, k = i - b * K;
    int s = i;
    if (obj_mask[i] == 0) {
      int J = 0;
      for (int q = 0; q <= b; ++q) J += nobj[q];  // objects in scenes 0..b
      const int rank = k - pre[i];                // background slots before this one in its scene
      if (use && total > 0 && rank < total - nobj[b]) s = pool[(J + rank) % total];
    }
    src[i] = s;
  }
}

// out[i,:] = x[src[i],:]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, const int *__restrict__ src, long long R,
                                                          int D, float *__restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int q = D / 4;
  if (t >= R * q) return;
  const long long i = t / q;
  const int c = (int)(t - i * q) * 4;
  *reinterpret_cast<float4 *>(out + i * D + c) = *reinterpret_cast<const float4 *>(x + (long long)src[i] * D + c);
}
// dx[src[i],:] += g[i,:]   (dx zeroed by the caller; several proposals may copy the same object row)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float *__restrict__ g, const int *__restrict__ src, long long R,
                                                           int D, float *__restrict__ dx) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= R * D) return;
  const long long i = t / D;
  const int c = (int)(t - i * D);
  atomicAdd(dx + (long long)src[i] * D + c, g[i * D + c]);
}

unsigned blocks_of(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// ---- answer loss (lib/loss_helper/loss_answering.py:11-13): sum of binary_cross_entropy_with_logits(x, t) / rows --------
// fwd: per-workgroup fp64 partials -> partial[blockIdx.x]; bwd: dx = g * (sigmoid(x) - t) / rows.  The stable form
// max(x, 0) - x t + log1p(exp(-|x|)) is torch's.
__global__ __launch_bounds__(256) void bce_logits_fwd_kernel(const float *__restrict__ x, const float *__restrict__ t, long long n,
                                                             double *__restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float xv = x[i];
    s += (double)(fmaxf(xv, 0.f) - xv * t[i] + log1pf(__expf(-fabsf(xv))));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void bce_logits_sum_kernel(const double *__restrict__ partial, int nblk, double inv_rows,
                                                            float *__restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (threadIdx.x == 0) out[0] = (float)(s * inv_rows);
}
__global__ __launch_bounds__(256) void bce_logits_bwd_kernel(const float *__restrict__ x, const float *__restrict__ t, long long n,
                                                             const float *__restrict__ g, float inv_rows, float *__restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float xv = x[i];
  dx[i] = g[0] * inv_rows * (1.0f / (1.0f + __expf(-xv)) - t[i]);
}

// Tail of get_joint_loss (loss_joint.py:204-223): the weighted total of the fused core + the optional terms, in the reference's
// fp32 order — ((core + w_lang lang) + (w_lcon lcon + w_icon icon)) + answer + caption — and the contrastive sum it reports.
__global__ void loss_tail_fwd_kernel(const float *__restrict__ core, const float *__restrict__ lang, const float *__restrict__ lcon,
                                     const float *__restrict__ icon, const float *__restrict__ ans, const float *__restrict__ cap,
                                     float w_lang, float w_lcon, float w_icon, float *__restrict__ total) {
  if (threadIdx.x != 0) return;
  float loss = core[0];
  if (lang) loss = loss + w_lang * lang[0];
  float con = 0.f;
  if (lcon && icon) {
    con = w_lcon * lcon[0] + w_icon * icon[0];
    loss = loss + con;
  }
  if (ans) loss = loss + ans[0];
  if (cap) loss = loss + cap[0];
  total[0] = loss;
  total[1] = con;
}
// d[0..n) = g e_at (the core's vector of reported scalars: only its total carries gradient), then [g, w_lang g, w_lcon g, w_icon g]
__global__ void loss_tail_bwd_kernel(const float *__restrict__ g, int n, int at, float w_lang, float w_lcon, float w_icon,
                                     float *__restrict__ d) {
  const float gv = g[0];
  for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = i == at ? gv : 0.f;
  if (threadIdx.x == 0) {
    d[n] = gv;
    d[n + 1] = w_lang * gv;
    d[n + 2] = w_lcon * gv;
    d[n + 3] = w_icon * gv;
  }
}

}  // namespace

extern "C" int vlp3d_roi_split(const float *out, int ld, long long R, int NH, int NC, float res_scale, float *hreg, float *hres,
                               float *hcls, float *rois, float *obj, float *sem, long long *obj_mask, long long *sem_arg,
                               void *stream) {
  if (!out || !hreg || !hres || !hcls || !rois || !obj || !sem || !obj_mask || !sem_arg || R < 1 || NH < 1 || NC < 1 ||
      ld < 2 * NH + 8 + NC)
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(roi_split_kernel, dim3(blocks_of(R, 256)), dim3(256), 0, (hipStream_t)stream, out, ld, R, NH, NC, res_scale,
                     hreg, hres, hcls, rois, obj, sem, obj_mask, sem_arg);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_roi_split_bwd(const float *d_hreg, const float *d_hres, const float *d_hcls, const float *d_rois,
                                   const float *d_obj, const float *d_sem, const float *rois, long long R, int NH, int NC,
                                   float res_scale, float *d_out, int ld, void *stream) {
  if (!rois || !d_out || R < 1 || NH < 1 || NC < 1 || ld < 2 * NH + 8 + NC) return VLP3D_EINVAL;
  hipLaunchKernelGGL(roi_split_bwd_kernel, dim3(blocks_of(R * ld, 256)), dim3(256), 0, (hipStream_t)stream, d_hreg, d_hres, d_hcls,
                     d_rois, d_obj, d_sem, rois, R, NH, NC, res_scale, d_out, ld);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_vote_epilogue(const float *seed_xyz, const float *seed_f, const float *net, int ld, long long R, int C,
                                   float *vote_xyz, float *vote_f, float *norm, void *stream) {
  if (!seed_xyz || !seed_f || !net || !vote_xyz || !vote_f || !norm || R < 1 || C < 1 || ld < 3 + C) return VLP3D_EINVAL;
  hipLaunchKernelGGL(vote_epilogue_kernel, dim3(blocks_of(R, 4)), dim3(256), 0, (hipStream_t)stream, seed_xyz, seed_f, net, ld, R,
                     C, vote_xyz, vote_f, norm);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_vote_epilogue_bwd(const float *d_vote_xyz, const float *d_vote_f, const float *vote_f, const float *norm,
                                       long long R, int C, float *d_seed_f, float *d_net, int ld, void *stream) {
  if (!vote_f || !norm || !d_seed_f || !d_net || R < 1 || C < 1 || ld < 3 + C) return VLP3D_EINVAL;
  hipLaunchKernelGGL(vote_epilogue_bwd_kernel, dim3(blocks_of(R, 4)), dim3(256), 0, (hipStream_t)stream, d_vote_xyz, d_vote_f,
                     vote_f, norm, R, C, d_seed_f, d_net, ld);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_l2norm_rows(const float *x, long long R, int C, float eps, float *y, float *norm, void *stream) {
  if (!x || !y || !norm || R < 1 || C < 1) return VLP3D_EINVAL;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(blocks_of(R, 4)), dim3(256), 0, (hipStream_t)stream, x, R, C, eps, y, norm);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_l2norm_rows_bwd(const float *g, const float *y, const float *norm, long long R, int C, float eps, float *dx,
                                     void *stream) {
  if (!g || !y || !norm || !dx || R < 1 || C < 1) return VLP3D_EINVAL;
  hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(blocks_of(R, 4)), dim3(256), 0, (hipStream_t)stream, g, y, norm, R, C, eps, dx);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_relation_inputs(const float *pc, int Cpc, int col0, int N, const int *seed_inds, int S, const int *vote_inds,
                                     const float *corners, int B, int K, float *obj_feat, float *bbox_feat, float *centre,
                                     void *stream) {
  if (!pc || !seed_inds || !vote_inds || !corners || !obj_feat || !bbox_feat || !centre || B < 1 || K < 1 || N < 1 || S < 1 ||
      col0 < 0 || Cpc < col0 + 128)
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(relation_inputs_kernel<false>, dim3(blocks_of((long long)B * K, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const void *)pc, Cpc, col0, N, seed_inds, S, vote_inds, corners, B, K, obj_feat, bbox_feat, centre);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
// the same with pc as bf16 rows (Cpc = their row stride in elements): obj_feat = the widened values
extern "C" int vlp3d_relation_inputs_bf16(const void *pc, int Cpc, int col0, int N, const int *seed_inds, int S,
                                          const int *vote_inds, const float *corners, int B, int K, float *obj_feat,
                                          float *bbox_feat, float *centre, void *stream) {
  if (!pc || !seed_inds || !vote_inds || !corners || !obj_feat || !bbox_feat || !centre || B < 1 || K < 1 || N < 1 || S < 1 ||
      col0 < 0 || Cpc < col0 + 128)
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(relation_inputs_kernel<true>, dim3(blocks_of((long long)B * K, 4)), dim3(256), 0, (hipStream_t)stream, pc, Cpc,
                     col0, N, seed_inds, S, vote_inds, corners, B, K, obj_feat, bbox_feat, centre);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_copy_paste_map(const long long *obj_mask, int B, int K, const float *coin, int *src, void *stream) {
  if (!obj_mask || !coin || !src || B < 1 || B > 1024 || K < 1 || (long long)B * K > 8192) return VLP3D_EINVAL;
  const size_t lds = ((size_t)2 * B * K + 2 * B) * sizeof(int);
  hipLaunchKernelGGL(copy_paste_map_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, obj_mask, B, K, coin, src);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_gather_rows(const float *x, const int *src, long long R, int D, float *out, void *stream) {
  if (!x || !src || !out || R < 1 || D < 4 || (D & 3)) return VLP3D_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_of(R * (D / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, src, R, D, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_scatter_rows_add(const float *g, const int *src, long long R, int D, float *dx, void *stream) {
  if (!g || !src || !dx || R < 1 || D < 1) return VLP3D_EINVAL;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks_of(R * D, 256)), dim3(256), 0, (hipStream_t)stream, g, src, R, D, dx);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// ---- AdamW on ONE flat parameter / gradient buffer ---------------------------------------------------------------------
// torch.optim.AdamW's update (decoupled weight decay, bias correction) for every element whose parameter took part in the
// step (active[seg[i]] != 0; parameters without gradient are skipped like torch skips grad-None parameters):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / c1) * m / (sqrt(v) / sqrt(c2) + eps)
// with c1 = 1 - b1^t, c2 = 1 - b2^t computed on the host.  One launch instead of the multi-tensor optimiser's six.
namespace {
__global__ __launch_bounds__(256) void adamw_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, const unsigned char *__restrict__ active,
                                                         long long n, float lr, float b1, float b2, float eps, float wd,
                                                         float c1, float sqrt_c2) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !active[i]) return;
  const float gi = g[i];
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  pi -= (lr / c1) * (mi / (sqrtf(vi) / sqrt_c2 + eps));
  p[i] = pi;
}
}  // namespace

extern "C" int vlp3d_adamw_flat(float *p, const float *g, float *m, float *v, const unsigned char *active, long long n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, float bias_c1, float sqrt_bias_c2,
                                void *stream) {
  if (!p || !g || !m || !v || !active || n < 1) return VLP3D_EINVAL;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, active,
                     n, lr, beta1, beta2, eps, weight_decay, bias_c1, sqrt_bias_c2);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// ---- copy_batch ------------------------------------------------------------------------------------------------------
// Many small device-to-device copies in ONE launch (the hand-over of the prepared backbone geometry is 16 index /
// coordinate tensors: torch._foreach_copy_ turned them into 16 memcpy nodes + 2 kernels per hand-over).  The table
// travels by value in the kernel arguments; a block moves one 16 KB piece of one entry.
namespace {
constexpr int COPY_BATCH = 48;
constexpr int COPY_PIECE = 16384;
struct CopyBatch {
  vlp3d_copy_desc d[COPY_BATCH];
  int first_block[COPY_BATCH + 1];
  int count;
};
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatch t) {
  int e = 0;
  while (e + 1 < t.count && (int)blockIdx.x >= t.first_block[e + 1]) ++e;
  const vlp3d_copy_desc &d = t.d[e];
  const long long off = (long long)((int)blockIdx.x - t.first_block[e]) * COPY_PIECE;
  const long long n = d.bytes - off < COPY_PIECE ? d.bytes - off : COPY_PIECE;
  const char *s = (const char *)d.src + off;
  char *o = (char *)d.dst + off;
  if ((((size_t)s | (size_t)o) & 15) == 0) {
    const long long nv = n >> 4;
    for (long long i = threadIdx.x; i < nv; i += 256) reinterpret_cast<uint4 *>(o)[i] = reinterpret_cast<const uint4 *>(s)[i];
    for (long long i = (nv << 4) + threadIdx.x; i < n; i += 256) o[i] = s[i];
  } else {
    for (long long i = threadIdx.x; i < n; i += 256) o[i] = s[i];
  }
}
}  // namespace

// ---- K-major copies of many small weight matrices in one launch (3dvlp_amd/row_mlp.py: prepared_weights) ---------------
// dst (cols x ld_dst) = src (rows x cols)^T for every job; one workgroup = one 32 x 32 tile through LDS (coalesced both ways).
namespace {
constexpr int TR_BATCH = 48;
struct TransposeBatch {
  vlp3d_transpose_desc d[TR_BATCH];
  int first_block[TR_BATCH + 1];
  int count;
};
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch t) {
  __shared__ float tile[32][33];
  int e = 0;
  while (e + 1 < t.count && (int)blockIdx.x >= t.first_block[e + 1]) ++e;
  const vlp3d_transpose_desc &d = t.d[e];
  const int tiles_c = (d.cols + 31) / 32;
  const int tb = (int)blockIdx.x - t.first_block[e];
  const int r0 = (tb / tiles_c) * 32, c0 = (tb % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float *src = (const float *)d.src;
  float *dst = (float *)d.dst;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < d.rows && c < d.cols) ? src[(long long)r * d.cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < d.cols && r < d.rows) dst[(long long)c * d.ld_dst + r] = tile[tx][i];
  }
}
}  // namespace

extern "C" int vlp3d_transpose_batch(const vlp3d_transpose_desc *descs, int count, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return VLP3D_EINVAL;
  for (int c0 = 0; c0 < count; c0 += TR_BATCH) {
    TransposeBatch t = {};
    t.count = count - c0 < TR_BATCH ? count - c0 : TR_BATCH;
    long long blocks = 0;
    for (int j = 0; j < t.count; ++j) {
      const vlp3d_transpose_desc &d = descs[c0 + j];
      if (!d.src || !d.dst || d.rows < 1 || d.cols < 1 || d.ld_dst < d.rows) return VLP3D_EINVAL;
      t.d[j] = d;
      t.first_block[j] = (int)blocks;
      blocks += (long long)((d.rows + 31) / 32) * ((d.cols + 31) / 32);
      if (blocks >= (1ll << 31)) return VLP3D_EINVAL;
    }
    t.first_block[t.count] = (int)blocks;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    VLP3D_LAUNCH_CHECK();
  }
  return VLP3D_OK;
}

extern "C" int vlp3d_copy_batch(const vlp3d_copy_desc *descs, int count, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return VLP3D_EINVAL;
  for (int c0 = 0; c0 < count; c0 += COPY_BATCH) {
    CopyBatch t = {};
    t.count = count - c0 < COPY_BATCH ? count - c0 : COPY_BATCH;
    long long blocks = 0;
    for (int j = 0; j < t.count; ++j) {
      const vlp3d_copy_desc &d = descs[c0 + j];
      if (d.bytes < 0 || (d.bytes > 0 && (!d.src || !d.dst))) return VLP3D_EINVAL;
      t.d[j] = d;
      t.first_block[j] = (int)blocks;
      blocks += (d.bytes + COPY_PIECE - 1) / COPY_PIECE;
      if (blocks >= (1ll << 31)) return VLP3D_EINVAL;
    }
    t.first_block[t.count] = (int)blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    VLP3D_LAUNCH_CHECK();
  }
  return VLP3D_OK;
}

// ---- small linear layers the MFMA kernels do not cover ----------------------------------------------------------------
// smallk: K <= 32 input columns (relation_module.py:64 bbox_embedding: Linear(27, 128) on 2048 proposals) — the BLAS
//         library spent 10 us forward and 33 + 8 us on the weight / bias gradients.
// rowdot: one output column (match_module.py:47: Linear(128, 1) on 16384 rows) — 10 us forward, 53 + 9 + 6 us backward.
namespace {
constexpr int SMALLK_KP = 32;

// out[r][n] = (base ? base[r][n] : 0) + x[r][:K] . W[n][:K] + b[n];  block = 16 rows x N columns (N in {64,128,256})
// (N is a template parameter: with a run-time rows_per the sixteen predicated multiply-adds of a k step each sat behind their
// own branch and LDS wait — 0.35 us per k, 14.9 us for 2048 x 27 -> 128 whatever R)
template <int N>
__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ W,
                                                         const float *__restrict__ bias, const float *__restrict__ base,
                                                         long long R, int K, float *__restrict__ out) {
  extern __shared__ int sm[];  // (one dynamic-LDS symbol per translation unit: declared int[] above)
  const int NP = N + 1;                       // (padded: the transposing writes below walk k fastest — stride N would be one bank)
  float *Wt = reinterpret_cast<float *>(sm);  // [K][NP]
  float *xs = Wt + (size_t)K * NP;            // [16][SMALLK_KP]
  const long long r0 = (long long)blockIdx.x * 16;
  {  // the weight (N x K, K <= 32) read as ONE coalesced stream — element e = n * K + k, 16 per thread in flight — and
    // transposed on its way into LDS.  (A thread per weight row read 32 dwords 4 K bytes apart: every load instruction of a
    // wave touched ~54 cache lines, and only N of the 256 threads took part: 14.7 us in-step for 2048 x 27 -> 128.)
    const int total = N * K;  // <= 256 * 32
    const unsigned long long kinv = ((1ull << 32) + K - 1) / K;  // e / K == (e * kinv) >> 32 for e * K < 2^32
    float wv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) wv[u] = W[min((int)threadIdx.x + 256 * u, total - 1)];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int e = threadIdx.x + 256 * u;
      if (e < total) {
        const int n = (int)(((unsigned long long)e * kinv) >> 32), k = e - n * K;
        Wt[k * NP + n] = wv[u];
      }
    }
  }
  for (int e = threadIdx.x; e < 16 * SMALLK_KP; e += 256) {
    const int rr = e / SMALLK_KP, k = e - rr * SMALLK_KP;
    xs[e] = (k < K && r0 + rr < R) ? x[(r0 + rr) * ldx + k] : 0.f;
  }
  __syncthreads();
  constexpr int groups = 256 / N, rows_per = 16 / groups;  // N = 64: 4 groups x 4 rows; 128: 2 x 8; 256: 1 x 16
  const int n = threadIdx.x % N, rg = threadIdx.x / N;
  float acc[rows_per], bs[rows_per];
#pragma unroll
  for (int j = 0; j < rows_per; ++j) {  // the residual rows are requested before the products, not behind them
    acc[j] = 0.f;
    const long long r = min(r0 + rg * rows_per + j, R - 1);
    bs[j] = base ? base[r * N + n] : 0.f;
  }
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    const float wv = Wt[k * NP + n];
#pragma unroll
    for (int j = 0; j < rows_per; ++j) acc[j] = __builtin_fmaf(xs[(rg * rows_per + j) * SMALLK_KP + k], wv, acc[j]);
  }
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int j = 0; j < rows_per; ++j) {
    const long long r = r0 + rg * rows_per + j;
    if (r < R) out[r * N + n] = bs[j] + acc[j] + bv;
  }
}

// slab[blockIdx.x] = [N x 32 partial dW (columns >= K zero) | N partial db] over this block's 64 rows
__global__ __launch_bounds__(256) void smallk_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x, int ldx,
                                                         long long R, int K, int N, float *__restrict__ slabs) {
  __shared__ float xs[64 * SMALLK_KP];
  extern __shared__ int sm[];
  float *dys = reinterpret_cast<float *>(sm);  // [64][N]: the block's dY rows, staged with all loads in flight (the loop
  //                                              below used to wait for one global load per row: 64 dependent round trips)
  const long long r0 = (long long)blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * SMALLK_KP; e += 256) {
    const int rr = e / SMALLK_KP, k = e - rr * SMALLK_KP;
    xs[e] = (k < K && r0 + rr < R) ? x[(r0 + rr) * ldx + k] : 0.f;
  }
  for (int e = threadIdx.x; e < 64 * (N / 4); e += 256) {
    const int rr = e / (N / 4), c4 = e - rr * (N / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + rr < R) v = *reinterpret_cast<const float4 *>(dy + (r0 + rr) * N + 4 * c4);
    *reinterpret_cast<float4 *>(dys + rr * N + 4 * c4) = v;
  }
  __syncthreads();
  float *slab = slabs + (size_t)blockIdx.x * ((size_t)N * SMALLK_KP + N);
  const int per = 256 / 2;  // threads per k-half
  const int kh = threadIdx.x / per;
  for (int n = threadIdx.x % per; n < N; n += per) {
    float acc[16], db = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int rr = 0; rr < 64; ++rr) {
      const float d = dys[rr * N + n];
      db += d;
      const float4 *xp = reinterpret_cast<const float4 *>(xs + rr * SMALLK_KP + 16 * kh);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = xp[q];
        acc[4 * q] = __builtin_fmaf(d, v.x, acc[4 * q]);
        acc[4 * q + 1] = __builtin_fmaf(d, v.y, acc[4 * q + 1]);
        acc[4 * q + 2] = __builtin_fmaf(d, v.z, acc[4 * q + 2]);
        acc[4 * q + 3] = __builtin_fmaf(d, v.w, acc[4 * q + 3]);
      }
    }
    float4 *o = reinterpret_cast<float4 *>(slab + (size_t)n * SMALLK_KP + 16 * kh);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    if (kh == 0) slab[(size_t)N * SMALLK_KP + n] = db;
  }
}

// y[r] = x[r][:K] . w + b: half a wave per row (32 lanes x float4 = 128 columns per pass)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ b, long long R, int K,
                                                         float *__restrict__ y) {
  const long long r = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int c = threadIdx.x & 31;
  float s = 0.f;
  if (r < R)
    for (int k = 4 * c; k < K; k += 128) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + r * K + k), wv = *reinterpret_cast<const float4 *>(w + k);
      s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
    }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (r < R && c == 0) y[r] = s + (b ? b[0] : 0.f);
}

// dx[r][k] = dy[r] w[k];  slab[blockIdx.x] = [partial dw (K) | partial db, 0, 0, 0] over the block's rows (K <= 128)
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                         const float *__restrict__ w, long long R, int K, int rows_per_block,
                                                         float *__restrict__ dx, float *__restrict__ slabs) {
  __shared__ float4 red[8][32];
  __shared__ float redb[8];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const bool on = 4 * c < K;
  const float4 wv = on ? *reinterpret_cast<const float4 *>(w + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float db = 0.f;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  for (int i = rg; i < rows_per_block; i += 8) {
    const long long r = r0 + i;
    if (r >= R) break;
    const float d = dy[r];
    db += d;
    if (on) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + r * K + 4 * c);
      acc.x = __builtin_fmaf(d, xv.x, acc.x); acc.y = __builtin_fmaf(d, xv.y, acc.y);
      acc.z = __builtin_fmaf(d, xv.z, acc.z); acc.w = __builtin_fmaf(d, xv.w, acc.w);
      if (dx) *reinterpret_cast<float4 *>(dx + r * K + 4 * c) = make_float4(d * wv.x, d * wv.y, d * wv.z, d * wv.w);
    }
  }
  red[rg][c] = acc;
  if (c == 0) redb[rg] = db;
  __syncthreads();
  if (rg == 0) {
    float4 t = red[0][c];
    float tb = redb[0];
#pragma unroll
    for (int g = 1; g < 8; ++g) {
      const float4 v = red[g][c];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      tb += redb[g];
    }
    float *slab = slabs + (size_t)blockIdx.x * (K + 4);
    if (on) *reinterpret_cast<float4 *>(slab + 4 * c) = t;
    if (c == 0) *reinterpret_cast<float4 *>(slab + K) = make_float4(tb, 0.f, 0.f, 0.f);
  }
}
}  // namespace

extern "C" int vlp3d_smallk_fwd(const float *x, int ldx, const float *W, const float *bias, const float *base, long long R,
                                int K, int N, float *out, void *stream) {
  if (!x || !W || !out || R < 1 || K < 1 || K > SMALLK_KP || ldx < K || (N != 64 && N != 128 && N != 256)) return VLP3D_EINVAL;
  const size_t lds = ((size_t)K * (N + 1) + 16 * SMALLK_KP) * sizeof(float);
  const dim3 grid((unsigned)((R + 15) / 16));
  if (N == 64) hipLaunchKernelGGL(smallk_fwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, x, ldx, W, bias, base, R, K, out);
  else if (N == 128) hipLaunchKernelGGL(smallk_fwd_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, x, ldx, W, bias, base, R, K, out);
  else hipLaunchKernelGGL(smallk_fwd_kernel<256>, grid, dim3(256), lds, (hipStream_t)stream, x, ldx, W, bias, base, R, K, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

/* slabs: ceil(R/64) x (N*32 + N) floats; sum them with vlp3d_slab_reduce_batch {nblk = ceil(R/64), n_mat = N*32, K = 32,
 * ldo = K_true, ncol_out = K_true, n_bias = N} */
extern "C" int vlp3d_smallk_bwd(const float *dy, const float *x, int ldx, long long R, int K, int N, float *slabs, void *stream) {
  if (!dy || !x || !slabs || R < 1 || K < 1 || K > SMALLK_KP || ldx < K || N < 4 || (N & 3)) return VLP3D_EINVAL;
  const size_t lds = (size_t)64 * N * sizeof(float);  // + 8 KB static: above 64 KB per workgroup at N = 256
  if (lds > 96 * 1024) return VLP3D_EINVAL;
  if (lds + 64 * SMALLK_KP * sizeof(float) > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};
    const int e = vlp3d_opt_in_lds(reinterpret_cast<const void *>(smallk_bwd_kernel), 96 * 1024, done);
    if (e != VLP3D_OK) return e;
  }
  hipLaunchKernelGGL(smallk_bwd_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), lds, (hipStream_t)stream, dy, x, ldx, R, K, N,
                     slabs);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_rowdot_fwd(const float *x, const float *w, const float *b, long long R, int K, float *y, void *stream) {
  if (!x || !w || !y || R < 1 || K < 4 || (K & 3)) return VLP3D_EINVAL;
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((unsigned)((R * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, R, K, y);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

/* slabs: nblk x (K + 4) floats, nblk = ceil(R / rows_per_block); sum with vlp3d_slab_reduce_batch {n_mat = K + 4, K = K + 4,
 * ldo = K + 4}: [dw (K) | db, 0, 0, 0] */
extern "C" int vlp3d_rowdot_bwd(const float *dy, const float *x, const float *w, long long R, int K, int rows_per_block,
                                float *dx, float *slabs, void *stream) {
  if (!dy || !x || !w || !slabs || R < 1 || K < 4 || (K & 3) || K > 128 || rows_per_block < 8) return VLP3D_EINVAL;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((unsigned)((R + rows_per_block - 1) / rows_per_block)), dim3(256), 0,
                     (hipStream_t)stream, dy, x, w, R, K, rows_per_block, dx, slabs);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// sum(binary_cross_entropy_with_logits(x, t)) / rows over (rows, cols) logits / soft targets (loss_answering.py:11-13).
// partial: (vlp3d_bce_logits_blocks(n)) doubles of scratch; bwd: g = dLoss (1 float on the device).
extern "C" int vlp3d_bce_logits_blocks(long long n) {
  const long long b = (n + 2047) / 2048;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
extern "C" int vlp3d_bce_logits_fwd(const float *x, const float *t, long long rows, long long cols, double *partial, float *out,
                                    void *stream) {
  if (!x || !t || !partial || !out || rows < 1 || cols < 1) return VLP3D_EINVAL;
  const int nblk = vlp3d_bce_logits_blocks(rows * cols);
  hipLaunchKernelGGL(bce_logits_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, t, rows * cols, partial);
  VLP3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(bce_logits_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, 1.0 / (double)rows, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
extern "C" int vlp3d_bce_logits_bwd(const float *x, const float *t, long long rows, long long cols, const float *g, float *dx,
                                    void *stream) {
  if (!x || !t || !g || !dx || rows < 1 || cols < 1) return VLP3D_EINVAL;
  const long long n = rows * cols;
  hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, t, n, g,
                     1.0f / (float)rows, dx);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// total[0] = ((core[0] + w_lang lang[0]) + (w_lcon lcon[0] + w_icon icon[0])) + ans[0] + cap[0], total[1] = the contrastive sum;
// lang / (lcon, icon) / ans / cap may be NULL (term absent).  One launch instead of the op-by-op scalar arithmetic.
extern "C" int vlp3d_loss_tail_fwd(const float *core, const float *lang, const float *lcon, const float *icon, const float *ans,
                                   const float *cap, float w_lang, float w_lcon, float w_icon, float *total, void *stream) {
  if (!core || !total || ((lcon == nullptr) != (icon == nullptr))) return VLP3D_EINVAL;
  hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, core, lang, lcon, icon, ans, cap, w_lang,
                     w_lcon, w_icon, total);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
// d (n + 4): d[0..n) = g[0] e_at (cotangent of the core's n reported scalars, total at index `at`), d[n..] = g[0] * [1, w_lang,
// w_lcon, w_icon] (answer / caption, language, the two contrastive terms).
extern "C" int vlp3d_loss_tail_bwd(const float *g, int n, int at, float w_lang, float w_lcon, float w_icon, float *d, void *stream) {
  if (!g || !d || n < 1 || at < 0 || at >= n) return VLP3D_EINVAL;
  hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, n, at, w_lang, w_lcon, w_icon, d);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
