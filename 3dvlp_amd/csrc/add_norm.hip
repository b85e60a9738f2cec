// out = LayerNorm(x + dropout(y)) — the "add & norm" that closes every attention / FFN block of the reference
// (models/transformer/attention.py:128-130 `self.layer_norm(queries + self.dropout(out))`, mmattention.py:84-86
// `self.norm(self.dropout(self.ffn(x)) + x)`) — as ONE kernel forward and one (+ a tiny slab sum) backward, instead
// of dropout / add / layer-norm forward and five-six launches in autograd's backward.
//
// One wave per row (D = 64*EPL columns, EPL consecutive columns per lane -> coalesced 8/16-byte accesses), row
// statistics by DPP-free shuffles in fp32.  The dropout mask is never stored: it is a counter-based hash of
// (seed, call id, element index), recomputed bit-for-bit in the backward kernel.  `seed` lives in device memory
// (one 64-bit word the step driver advances once per step, inside the replayed graph), `call_id` distinguishes the
// call sites of a step.  Kept for backward: xhat (the normalised rows) and rstd.
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int EPL>
__global__ __launch_bounds__(256) void add_norm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, long long R, float p,
                                                           const unsigned long long *__restrict__ seed, int call_id,
                                                           float eps, float *__restrict__ out,
                                                           float *__restrict__ xhat, float *__restrict__ rstd,
                                                           unsigned char *__restrict__ mask, int std_mode,
                                                           float *__restrict__ sum_out, float *__restrict__ kappa, int rep,
                                                           int seq) {
  constexpr int D = 64 * EPL;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  // rep > 1: x and y hold ONE copy of every group of `seq` rows, the R output rows are `rep` consecutive copies of each
  // group (output row (g, l, k) reads source row (g, k)); the dropout mask is drawn per OUTPUT element
  const long long srow = rep > 1 ? (row / ((long long)rep * seq)) * seq + row % seq : row;
  const int c0 = lane * EPL;
  const unsigned thresh = (unsigned)(p * 16777216.0f);
  const unsigned mix = p > 0.f ? seed_mix_of(seed, call_id) : 0u;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float r[EPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    const long long e = row * D + c0 + i, es = srow * D + c0 + i;
    float yv = y ? y[es] : 0.f;
    if (p > 0.f) {
      const bool keep = keep_element(mix, (unsigned)e, thresh);
      yv = keep ? yv * inv_keep : 0.f;
      if (mask) mask[e] = keep ? 1 : 0;
    }
    r[i] = x[es] + yv;
    if (sum_out) sum_out[e] = r[i];
    s += r[i];
  }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    const float d = r[i] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  if (std_mode) {
    // the captioner's own LayerNorm (transformer_captioner.py:117-129): a * (x - mean) / (std + eps) + b with the UNBIASED
    // standard deviation and eps added to it.  With n = (x - mean) / (std + eps), r = 1 / (std + eps):
    //   dx = r * (g - mean(g) - n * mean(g n) * kappa),  kappa = D (std + eps) / ((D - 1) std)      (g = dout * a)
    // — the plain LayerNorm backward with one extra per-row factor, so both forms share the backward kernel.
    const float sd = sqrtf(q * (1.0f / (D - 1)));
    const float den = sd + eps;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const long long e = row * D + c0 + i;
      const float h = (r[i] - mean) / den;
      xhat[e] = h;
      out[e] = gamma[c0 + i] * h + beta[c0 + i];
    }
    if (lane == 0) {
      rstd[row] = 1.0f / den;
      kappa[row] = (float)D * den / ((float)(D - 1) * sd);
    }
    return;
  }
  const float rs = rsqrtf(q * (1.0f / D) + eps);
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    const long long e = row * D + c0 + i;
    const float h = (r[i] - mean) * rs;
    xhat[e] = h;
    out[e] = h * gamma[c0 + i] + beta[c0 + i];
  }
  if (lane == 0) {
    rstd[row] = rs;
    if (kappa) kappa[row] = 1.0f;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dout * gamma;   dy = dx * mask / (1 - p);
// per-workgroup partial sums of dgamma = sum_rows dout * xhat and dbeta = sum_rows dout go to slab [blockIdx][2][D].
template <int EPL>
__global__ __launch_bounds__(256) void add_norm_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ xhat,
                                                           const float *__restrict__ rstd,
                                                           const float *__restrict__ gamma, long long R, float p,
                                                           const unsigned long long *__restrict__ seed, int call_id,
                                                           int rows_per_wave, float *__restrict__ dx,
                                                           float *__restrict__ dy, float *__restrict__ partials,
                                                           const float *__restrict__ dres,
                                                           const float *__restrict__ kappa) {
  constexpr int D = 64 * EPL;
  __shared__ float red[4][2][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * EPL;
  const unsigned thresh = (unsigned)(p * 16777216.0f);
  const unsigned mix = p > 0.f ? seed_mix_of(seed, call_id) : 0u;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float gm[EPL], dg[EPL], db[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    gm[i] = gamma[c0 + i];
    dg[i] = db[i] = 0.f;
  }
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * rows_per_wave;
  // Four rows' operands are requested before the first row is reduced (clamped rows, no branch around a load): a wave used to
  // walk its rows one memory round trip at a time — eight dependent latencies, 26 us for 16 384 x 128 whatever the grid.  The
  // sums over rows keep their order.
  constexpr int RB = 4;
  const float *drp = dres ? dres : dout, *kpp = kappa ? kappa : rstd;  // (valid addresses for the unconditional loads)
  const bool has_dres = dres != nullptr, has_kappa = kappa != nullptr;
  for (int k0 = 0; k0 < rows_per_wave; k0 += RB) {
    if (row0 + k0 >= R) break;
    float dv[RB][EPL], hv[RB][EPL], rv[RB][EPL], rsv[RB], kpv[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const long long row = min(row0 + k0 + u, R - 1);
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const long long e = row * D + c0 + i;
        dv[u][i] = dout[e];
        hv[u][i] = xhat[e];
        rv[u][i] = drp[e];
      }
      rsv[u] = rstd[row];
      kpv[u] = kpp[row];
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const long long row = row0 + k0 + u;
      if (k0 + u >= rows_per_wave || row >= R) continue;  // (wave-uniform)
      float g[EPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const float d = dv[u][i];
        g[i] = d * gm[i];
        dg[i] += d * hv[u][i];
        db[i] += d;
        s1 += g[i];
        s2 += g[i] * hv[u][i];
      }
      const float m1 = wave_sum(s1) * (1.0f / D);
      float m2 = wave_sum(s2) * (1.0f / D);
      if (has_kappa) m2 *= kpv[u];
      const float rs = rsv[u];
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const long long e = row * D + c0 + i;
        float v = rs * (g[i] - m1 - hv[u][i] * m2);
        if (has_dres) v += rv[u][i];  // the sum x + dropout(y) was also an output (pre-norm residual stream): its gradient joins here
        dx[e] = v;
        if (dy) {
          float w = v;
          if (p > 0.f) w = keep_element(mix, (unsigned)e, thresh) ? v * inv_keep : 0.f;
          dy[e] = w;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    red[wave][0][c0 + i] = dg[i];
    red[wave][1][c0 + i] = db[i];
  }
  __syncthreads();
  float *slab = partials + (long long)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int which = i / D, c = i - which * D;
    slab[i] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// out[i] = sum_b partials[b][i]: a block sums 64 consecutive elements — 16 threads x float4 — in 16 slab-groups (each
// thread nblk/16 independent 16-byte loads), then folds the groups through LDS.  n % 4 == 0.
__global__ __launch_bounds__(256) void add_norm_slab_sum_kernel(const float *__restrict__ partials, int nblk, int n,
                                                                float *__restrict__ out) {
  __shared__ float4 red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 64 + 4 * q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
#pragma unroll 4
    for (int b = grp; b < nblk; b += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(partials + (long long)b * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[grp][q] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float4 t = red[0][q];
#pragma unroll
    for (int g = 1; g < 16; ++g) {
      const float4 v = red[g][q];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + i) = t;
  }
}

}  // namespace

// ---- activation + dropout (attention.py:104-112 PositionwiseFeedForward: dropout(relu(.)); match_module.py:40-47:
// Dropout(GELU(.))) as ONE element-wise kernel each way, with the same never-stored hash mask as add & norm.
// kind 0 = ReLU, 1 = GELU (erf form, torch's default).  Backward recomputes act'(z) and the mask from z.
__device__ __forceinline__ float act_fwd(float z, int kind) {
  return kind == 0 ? fmaxf(z, 0.f) : 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
}
__device__ __forceinline__ float act_grad(float z, int kind) {
  if (kind == 0) return z > 0.f ? 1.f : 0.f;
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  return cdf + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
}
__global__ __launch_bounds__(256) void act_dropout_kernel(const float *__restrict__ z, const float *__restrict__ dout,
                                                          long long n4, int kind, float p,
                                                          const unsigned long long *__restrict__ seed, int call_id,
                                                          float *__restrict__ out, unsigned char *__restrict__ mask) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const unsigned thresh = (unsigned)(p * 16777216.0f);
  const unsigned mix = p > 0.f ? seed_mix_of(seed, call_id) : 0u;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const float4 zv = reinterpret_cast<const float4 *>(z)[i];
  float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
  if (dout) g = reinterpret_cast<const float4 *>(dout)[i];
  const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, gg[4] = {g.x, g.y, g.z, g.w};
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool keep = p > 0.f ? keep_element(mix, (unsigned)(4 * i + j), thresh) : true;
    const float v = dout ? gg[j] * act_grad(zz[j], kind) : act_fwd(zz[j], kind);
    o[j] = keep ? v * inv_keep : 0.f;
    if (mask) mask[4 * i + j] = keep ? 1 : 0;
  }
  reinterpret_cast<float4 *>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
}

static int add_norm_rows_per_wave() {
  static const int v = getenv("VLP3D_ADDNORM_RPW") ? atoi(getenv("VLP3D_ADDNORM_RPW")) : 8;  // 16 left half of the CUs idle at 16384 rows
  return v < 1 ? 1 : v;
}
extern "C" int vlp3d_add_norm_blocks(long long R) {  // workgroups (= partial slabs) of the backward kernel
  const int rpw = add_norm_rows_per_wave();
  const long long waves = (R + rpw - 1) / rpw;
  long long blocks = (waves + 3) / 4;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static int add_norm_fwd_any(const float *x, const float *y, const float *gamma, const float *beta, long long R, int D,
                            float p, const unsigned long long *seed, int call_id, float eps, float *out, float *xhat,
                            float *rstd, unsigned char *mask, int std_mode, float *sum_out, float *kappa, void *stream,
                            int rep = 1, int seq = 0) {
  if (!x || !gamma || !beta || !out || !xhat || !rstd || R < 1 || p < 0.f || p >= 1.f || (p > 0.f && (!seed || !y)) ||
      (std_mode && !kappa) || R * (long long)D >= (1ll << 32))
    return VLP3D_EINVAL;
  const dim3 grid((unsigned)((R + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define ADD_NORM_FWD(EPL) \
  hipLaunchKernelGGL(add_norm_fwd_kernel<EPL>, grid, block, 0, s, x, y, gamma, beta, R, p, seed, call_id, eps, out, xhat, rstd, \
                     mask, std_mode, sum_out, kappa, rep, seq)
  switch (D) {
    case 64: ADD_NORM_FWD(1); break;
    case 128: ADD_NORM_FWD(2); break;
    case 256: ADD_NORM_FWD(4); break;
    default: return VLP3D_EINVAL;
  }
#undef ADD_NORM_FWD
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_add_norm_fwd(const float *x, const float *y, const float *gamma, const float *beta, long long R,
                                  int D, float p, const unsigned long long *seed, int call_id, float eps, float *out,
                                  float *xhat, float *rstd, unsigned char *mask, void *stream) {
  if (!y) return VLP3D_EINVAL;
  return add_norm_fwd_any(x, y, gamma, beta, R, D, p, seed, call_id, eps, out, xhat, rstd, mask, 0, nullptr, nullptr, stream);
}

// out (R rows) = LayerNorm(x + dropout_p(y)) where x, y hold R / rep rows: every group of `seq` source rows is used by `rep`
// consecutive output groups (match_module.py:127 tiles the proposals over the sentences BEFORE the first decoder layer,
// whose attention block — attention.py:41-78 has no dropout inside — is therefore identical for all copies; only this
// dropout + add & norm differs per copy).  R % (rep * seq) == 0.
extern "C" int vlp3d_add_norm_rep_fwd(const float *x, const float *y, const float *gamma, const float *beta, long long R,
                                      int D, int rep, int seq, float p, const unsigned long long *seed, int call_id, float eps,
                                      float *out, float *xhat, float *rstd, void *stream) {
  if (!y || rep < 1 || seq < 1 || R % ((long long)rep * seq)) return VLP3D_EINVAL;
  return add_norm_fwd_any(x, y, gamma, beta, R, D, p, seed, call_id, eps, out, xhat, rstd, nullptr, 0, nullptr, nullptr, stream,
                          rep, seq);
}

namespace {
// sx[g,k,:] = sum_l a[g,l,k,:], sy likewise for b: the adjoint of the row replication above, both tensors in one launch
__global__ __launch_bounds__(256) void rep_sum2_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, long long n4,
                                                       int rep, long long group4, float4 *__restrict__ sa, float4 *__restrict__ sb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over the (R / rep) * D / 4 output float4s
  if (i >= n4) return;
  const long long g = i / group4, k = i - g * group4;
  const float4 *pa = a + g * rep * group4 + k, *pb = b + g * rep * group4 + k;
  float4 ta = make_float4(0.f, 0.f, 0.f, 0.f), tb = ta;
  for (int l = 0; l < rep; ++l) {
    const float4 va = pa[(long long)l * group4], vb = pb[(long long)l * group4];
    ta.x += va.x; ta.y += va.y; ta.z += va.z; ta.w += va.w;
    tb.x += vb.x; tb.y += vb.y; tb.z += vb.z; tb.w += vb.w;
  }
  sa[i] = ta;
  sb[i] = tb;
}
}  // namespace

extern "C" int vlp3d_rep_sum2(const float *a, const float *b, long long rows_out, int D, int rep, int seq, float *sa, float *sb,
                              void *stream) {
  if (!a || !b || !sa || !sb || rows_out < 1 || D < 4 || (D & 3) || rep < 1 || seq < 1 || rows_out % seq) return VLP3D_EINVAL;
  const long long n4 = rows_out * D / 4, group4 = (long long)seq * D / 4;
  hipLaunchKernelGGL(rep_sum2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b), n4, rep, group4,
                     reinterpret_cast<float4 *>(sa), reinterpret_cast<float4 *>(sb));
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// Pre-norm residual stream (transformer_captioner.py:132-145 SublayerConnection: x + dropout(sublayer(norm(x)))): ONE
// launch produces the new stream value s = x + dropout_p(y) (sum_out; y == NULL: s = x, the first norm of a stack) AND
// norm(s) for the next sublayer.  std_mode 1 = the captioner's LayerNorm (:117-129, unbiased std, eps outside the root),
// 0 = nn.LayerNorm.  kappa (R) is scratch kept for backward (required with std_mode 1).
extern "C" int vlp3d_sum_norm_fwd(const float *x, const float *y, const float *gamma, const float *beta, long long R,
                                  int D, float p, const unsigned long long *seed, int call_id, float eps, int std_mode,
                                  float *sum_out, float *out, float *xhat, float *rstd, float *kappa,
                                  unsigned char *mask, void *stream) {
  return add_norm_fwd_any(x, y, gamma, beta, R, D, p, seed, call_id, eps, out, xhat, rstd, mask, std_mode ? 1 : 0, sum_out,
                          kappa, stream);
}

static int add_norm_bwd_any(const float *dout, const float *dres, const float *xhat, const float *rstd,
                            const float *kappa, const float *gamma, long long R, int D, float p,
                            const unsigned long long *seed, int call_id, float *dx, float *dy, float *partials,
                            float *dgamma_dbeta, int defer_reduce, void *stream) {
  if (!dout || !xhat || !rstd || !gamma || !dx || !partials || (!dgamma_dbeta && !defer_reduce) || R < 1 || p < 0.f ||
      p >= 1.f || (p > 0.f && !seed) || R * (long long)D >= (1ll << 32))
    return VLP3D_EINVAL;
  const int nblk = vlp3d_add_norm_blocks(R);
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
#define ADD_NORM_BWD(EPL) \
  hipLaunchKernelGGL(add_norm_bwd_kernel<EPL>, grid, block, 0, s, dout, xhat, rstd, gamma, R, p, seed, call_id, \
                     add_norm_rows_per_wave(), dx, dy, partials, dres, kappa)
  switch (D) {
    case 64: ADD_NORM_BWD(1); break;
    case 128: ADD_NORM_BWD(2); break;
    case 256: ADD_NORM_BWD(4); break;
    default: return VLP3D_EINVAL;
  }
#undef ADD_NORM_BWD
  if (!defer_reduce)  // else the caller sums the vlp3d_add_norm_blocks(R) slabs of 2*D floats later (vlp3d_slab_reduce_batch)
    hipLaunchKernelGGL(add_norm_slab_sum_kernel, dim3((2 * D + 63) / 64), dim3(256), 0, s, partials, nblk, 2 * D,
                       dgamma_dbeta);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_add_norm_bwd(const float *dout, const float *xhat, const float *rstd, const float *gamma,
                                  long long R, int D, float p, const unsigned long long *seed, int call_id, float *dx,
                                  float *dy, float *partials, float *dgamma_dbeta, void *stream) {
  if (!dy) return VLP3D_EINVAL;
  return add_norm_bwd_any(dout, nullptr, xhat, rstd, nullptr, gamma, R, D, p, seed, call_id, dx, dy, partials, dgamma_dbeta,
                          0, stream);
}

// Backward of vlp3d_sum_norm_fwd: dout = gradient of the normalised output, dres = gradient of sum_out (NULL: none);
// dx = total gradient of the stream value (= of x), dy = dx * mask / (1 - p) (NULL when there was no y).
// defer_reduce 1: [dgamma | dbeta] is NOT formed; the vlp3d_add_norm_blocks(R) slabs of 2*D floats in `partials` are left for
// vlp3d_slab_reduce_batch (n_mat = K = ldo = 2*D) — the step driver sums them with all the other slabs of the backward pass.
extern "C" int vlp3d_sum_norm_bwd(const float *dout, const float *dres, const float *xhat, const float *rstd,
                                  const float *kappa, const float *gamma, long long R, int D, float p,
                                  const unsigned long long *seed, int call_id, float *dx, float *dy, float *partials,
                                  float *dgamma_dbeta, int defer_reduce, void *stream) {
  return add_norm_bwd_any(dout, dres, xhat, rstd, kappa, gamma, R, D, p, seed, call_id, dx, dy, partials, dgamma_dbeta,
                          defer_reduce, stream);
}

// out = dropout_p(act(z)) (dout == NULL) or dz = dout * act'(z) * mask / (1-p) (dout given); n % 4 == 0, n < 2^32.
extern "C" int vlp3d_act_dropout(const float *z, const float *dout, long long n, int kind, float p,
                                 const unsigned long long *seed, int call_id, float *out, unsigned char *mask, void *stream) {
  if (!z || !out || n < 4 || (n & 3) || n >= (1ll << 32) || kind < 0 || kind > 1 || p < 0.f || p >= 1.f || (p > 0.f && !seed))
    return VLP3D_EINVAL;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(act_dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, dout, n4,
                     kind, p, seed, call_id, out, mask);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
