// Fused scaled-dot-product attention (forward + backward) for gfx950 — replaces the
// att = softmax(QK^T/sqrt(dk) [+bias | *w] [mask]) ; out = att V core of the reference's
// models/transformer/attention.py:63-75, which materialises att (b,h,nq,nk) in fp32 (67 MB per
// self-attention layer at the grounding shapes), re-reads it for softmax and AV and keeps it for
// backward.  Here nothing of size nq*nk touches memory (except the optional bias gradient).
//
// Shapes on the path: d_k = d_v = 32, h = 4, nq = 256, nk in {49, 256}: tiny per-head problems.
// One WAVE owns a 32-row tile of one (batch, head) and walks the other dimension in 32-wide
// tiles; all contractions run on the matrix cores with the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) so that the result stays within fp32 round-off of the reference.
// The products are arranged "transposed" (S^T = K Q^T, lane = query) so that
//   * the softmax row statistics are per-lane scalars (16 registers + one cross-half exchange),
//   * an accumulator tile is directly the B operand of the next product (no LDS, no shuffles):
//     MFMA step s of the second product sums over the row index that accumulator register s
//     holds, rho(s,half) = (s&3) + 8*(s>>2) + 4*half, and the A operand is read with that same
//     permutation of the summation index.
// The summation index of the first product is permuted too (half h of the wave covers head
// dims 16h..16h+15) so that every lane reads 64 contiguous bytes of its row.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 32;  // head dim (d_k == d_v)

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// row index held by accumulator register r of lane-half `half` (32x32 MFMA C/D layout)
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// 16 contiguous floats of a (.., H*D) row: elements [16*half, 16*half+16) of head `h`.
__device__ __forceinline__ void load_half_row(const float *__restrict__ base, long long row, int HD, int h, int half,
                                              float (&dst)[16]) {
  const float4 *p = reinterpret_cast<const float4 *>(base + row * HD + h * D + half * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 v = p[i];
    dst[4 * i + 0] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
  }
}

// C(32x32) += A(32x32) * B(32x32)^T-style product where lane (r = lane&31, half) supplies
// a[kk] = A[r][16*half+kk] and b[kk] = B[r][16*half+kk]:  C[i][j] = sum_d A[i][d] * B[j][d].
// BF = false: 16 exact-fp32 MFMA steps (v_mfma_f32_32x32x2_f32).  BF = true: the same contraction as two
// v_mfma_f32_32x32x16_bf16 — the operands are rounded to bf16 in registers, accumulation stays fp32 (1/16 of the
// matrix-core time; the timing configuration of the step, not the 1e-4 parity configuration).
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ short bf16_bits(float v) {
  __hip_bfloat16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}

// ---- operands that are ALREADY bf16 in memory (IO bits of vlp3d_sdpa_fwd_io / _bwd_io: 1 = q, 2 = k and v, 4 = out) ----
// The bf16-MFMA kernels round q / k / v to bf16 on their way into registers / LDS; when the producer (a row chain's last
// stage, a linear layer) stores them as bf16 rows the kernels move half the bytes and compute the same numbers.
__device__ __forceinline__ float bf16_to_f32(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }

// elements [16*half, 16*half + 16) of head h of a row as the two bf16 MFMA operands of a lane
template <bool B16>
__device__ __forceinline__ void load_half_row_bf(const void *__restrict__ base, long long row, int ld, int h, int half,
                                                 bf16x8 (&dst)[2]) {
  if (B16) {
    const bf16x8 *p = reinterpret_cast<const bf16x8 *>(reinterpret_cast<const short *>(base) + row * ld + h * D + half * 16);
    dst[0] = p[0];
    dst[1] = p[1];
  } else {
    float t[16];
    load_half_row(reinterpret_cast<const float *>(base), row, ld, h, half, t);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[u][j] = bf16_bits(t[8 * u + j]);
  }
}
// the same 16 elements as floats
template <bool B16>
__device__ __forceinline__ void load_half_row_f(const void *__restrict__ base, long long row, int ld, int h, int half,
                                                float (&dst)[16]) {
  if (B16) {
    bf16x8 t[2];
    load_half_row_bf<true>(base, row, ld, h, half, t);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[8 * u + j] = bf16_to_f32(t[u][j]);
  } else {
    load_half_row(reinterpret_cast<const float *>(base), row, ld, h, half, dst);
  }
}
// four consecutive elements at element offset `off` as bf16 bits
template <bool B16>
__device__ __forceinline__ short4 load4_bf(const void *__restrict__ base, long long off) {
  if (B16) return *reinterpret_cast<const short4 *>(reinterpret_cast<const short *>(base) + off);
  const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + off);
  short4 r;
  r.x = bf16_bits(v.x); r.y = bf16_bits(v.y); r.z = bf16_bits(v.z); r.w = bf16_bits(v.w);
  return r;
}

template <bool BF, typename VA, typename VB>
__device__ __forceinline__ f32x16 mfma_rows(const VA &a, const VB &b, f32x16 c) {
  if (BF) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 av, bv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        av[j] = bf16_bits(a[8 * t + j]);
        bv[j] = bf16_bits(b[8 * t + j]);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    }
    return c;
  }
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], c, 0, 0, 0);
  return c;
}

// bias_mode: 0 none, 1 add, 2 mul.  Returns the pre-softmax score of (query, key) given raw = q.k * scale.
__device__ __forceinline__ float apply_bias(float raw, int bias_mode, const float *__restrict__ bias,
                                            long long off) {
  if (bias_mode == 1) return raw + bias[off];
  if (bias_mode == 2) return raw * bias[off];
  return raw;
}

// The 16 bias values a lane (query row `bias_row`, half) needs for key tile k0: keys k0 + acc_row(i, half), i.e. four runs
// of four consecutive floats -> four 16-byte loads per lane when the row is 16-byte aligned and the tile is full (the
// element-wise form issued 16 dword loads per tile, each touching 32..64 cache lines of 32..64 different rows).
__device__ __forceinline__ void load_bias_tile(const float *__restrict__ bias, int bias_mode, long long bias_row, int k0,
                                               int half, int nk, float (&bt)[16]) {
  if (bias_mode == 0) return;
  if ((nk & 3) == 0 && k0 + 32 <= nk && (reinterpret_cast<size_t>(bias) & 15) == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4 *>(bias + bias_row + k0 + 8 * g + 4 * half);
      bt[4 * g + 0] = t.x; bt[4 * g + 1] = t.y; bt[4 * g + 2] = t.z; bt[4 * g + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, half);
      bt[i] = key < nk ? bias[bias_row + key] : 0.f;
    }
  }
}
__device__ __forceinline__ float apply_bias_value(float raw, int bias_mode, float bv) {
  return bias_mode == 1 ? raw + bv : (bias_mode == 2 ? raw * bv : raw);
}
__device__ __forceinline__ void store_dbias_tile(float *__restrict__ dbias, long long bias_row, int k0, int half, int nk,
                                                 const float (&dv)[16]) {
  if ((nk & 3) == 0 && k0 + 32 <= nk && (reinterpret_cast<size_t>(dbias) & 15) == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(dbias + bias_row + k0 + 8 * g + 4 * half) =
          make_float4(dv[4 * g + 0], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, half);
      if (key < nk) dbias[bias_row + key] = dv[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward: wave = 32 queries of one (b,h); loop over key tiles; online softmax.
// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(64) void sdpa_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ v, const float *__restrict__ bias,
                                                      int bias_mode, const float *__restrict__ mask, int H, int nq,
                                                      int nk, int ldq, int ldk, int ldv, float scale,
                                                      float *__restrict__ out, float *__restrict__ lse) {
  const int lane = threadIdx.x, r = lane & 31, half = lane >> 5;
  const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int HD = H * D;
  const int qi = min(q0 + r, nq - 1);
  const long long qrow = (long long)b * nq + qi;

  float qreg[16];
  load_half_row(q, qrow, ldq, h, half, qreg);

  f32x16 o = zero16();
  float m = -__builtin_inff(), l = 0.f;
  const long long bias_row = (((long long)b * H + h) * nq + qi) * nk;

  // software pipeline: the K rows and the V column of tile t+1 are requested before tile t is computed (a wave walks
  // its key tiles alone; without this every tile exposes two full memory latencies)
  float kreg[16], vcol[16];
  auto load_tile = [&](int k0, float (&kr)[16], float (&vc)[16]) {
    load_half_row(k, (long long)b * nk + min(k0 + r, nk - 1), ldk, h, half, kr);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = min(k0 + acc_row(i, half), nk - 1);  // p == 0 beyond nk
      vc[i] = v[((long long)b * nk + key) * ldv + h * D + r];
    }
  };
  load_tile(0, kreg, vcol);
  for (int k0 = 0; k0 < nk; k0 += 32) {
    float knext[16], vnext[16];
    load_tile(min(k0 + 32, nk - 1), knext, vnext);  // clamped: the last prefetch re-reads valid rows and is unused
    float bt[16];
    load_bias_tile(bias, bias_mode, bias_row, k0, half, nk, bt);
    f32x16 s = mfma_rows<BF>(kreg, qreg, zero16());  // s[reg] = S[query r][key k0 + acc_row(reg, half)]

    float tmax = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, half);
      float x = -__builtin_inff();
      if (key < nk) {
        x = apply_bias_value(s[i] * scale, bias_mode, bt[i]);
        if (mask != nullptr && mask[(long long)b * nk + key] == 0.f) x = -10000.f;
      }
      s[i] = x;
      tmax = fmaxf(tmax, x);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    const float alpha = __expf(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __expf(s[i] - m_new);
      s[i] = p;
      psum += p;
      o[i] *= alpha;
    }
    l = l * alpha + psum;
    m = m_new;
    // O^T[dim][query] += V^T[dim][key] * P^T[key][query]; step i sums over key k0 + acc_row(i, half)
    o = mfma_rows<BF>(vcol, s, o);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      kreg[i] = knext[i];
      vcol[i] = vnext[i];
    }
  }
  l += __shfl_xor(l, 32);
  if (q0 + r < nq) {
    const float inv = 1.f / l;
    float *__restrict__ orow = out + qrow * HD + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 = dims 8g + 4*half + 0..3
      float4 w;
      w.x = o[4 * g + 0] * inv; w.y = o[4 * g + 1] * inv; w.z = o[4 * g + 2] * inv; w.w = o[4 * g + 3] * inv;
      *reinterpret_cast<float4 *>(orow + 8 * g + 4 * half) = w;
    }
    if (half == 0) lse[((long long)b * H + h) * nq + q0 + r] = m + __logf(l);
  }
}

// Workgroup -> (block along x, head, batch element) for the LDS kernels.  The workgroups of one batch element share memory
// lines: the query blocks of a head stage the same K / V, and the heads of a row own 64-byte (bf16 rows) or 128-byte pieces of
// the same lines.  Workgroups go round-robin over the 8 XCDs by linear id, so with the launch's own (x, y, z) the sharers sit on
// different XCDs and every one of them pulls its lines through its own L2 (PMC: 21 MB fetched for 12.6 MB of bf16 q|k|v).
// Here the gridDim.x * gridDim.y workgroups of a batch element are given ids that are 8 apart: one XCD, dispatched together.
__device__ __forceinline__ void xcd_block(int &x, int &h, int &b) {
  const int nx = gridDim.x, H = gridDim.y, B = gridDim.z;
  if ((B & 7) == 0) {
    const int n = blockIdx.x + nx * (blockIdx.y + H * blockIdx.z);  // linear id = dispatch order
    const int per = nx * H, xcd = n & 7, s = n >> 3;
    const int local = s % per;
    b = (s / per) * 8 + xcd;
    x = local % nx;
    h = local / nx;
  } else {
    x = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
  }
}

// ---------------------------------------------------------------------------------------------
// forward, bf16-MFMA form with the head's K and V shared through LDS: a workgroup = 4 waves = 128 queries of one
// (b,h).  In the one-wave form every 32-query wave re-reads its head's K and V from L2 (PMC traffic 2.4x the
// algorithmic bytes, 20 dependent loads per key tile).  Here K and V are converted to bf16 ONCE while staging:
// K row-major [key][32 + 8] (a lane's MFMA operand = two ds_read_b128), V transposed [dim][nk + 4] (the operand of the
// second product = four ds_read_b64 in the accumulator's row order) — six LDS reads per key tile, no conversions.
// ---------------------------------------------------------------------------------------------
template <int IO>  // bit 0: q holds bf16, bit 1: k and v hold bf16, bit 2: out is written as bf16 (row strides in elements)
__global__ __launch_bounds__(256) void sdpa_fwd_lds_kernel(const void *__restrict__ q, const void *__restrict__ k,
                                                           const void *__restrict__ v, const float *__restrict__ bias,
                                                           int bias_mode, const float *__restrict__ mask, int H, int nq,
                                                           int nk, int nkp, int ldq, int ldk, int ldv, float scale,
                                                           void *__restrict__ out, float *__restrict__ lse) {
  constexpr bool Q16 = (IO & 1) != 0, KV16 = (IO & 2) != 0, O16 = (IO & 4) != 0;
  extern __shared__ __attribute__((aligned(16))) short sm_kv[];
  constexpr int KS = D + 8;        // K row stride (shorts): 80 B -> conflict-free b128 phases
  const int VS = nkp + 4;          // V^T row stride (shorts): (nkp/2 + 2) dwords -> conflict-free b64 phases
  short *sK = sm_kv;               // [nkp][KS]
  short *sV = sm_kv + nkp * KS;    // [D][VS]
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, h, b;
  xcd_block(bx, h, b);
  const int HD = H * D;
  // the wave's query rows are requested first: their latency runs under the staging of K and V
  const int q0 = (bx * 4 + wave) * 32;
  const int qi = min(q0 + r, nq - 1);
  const long long qrow = (long long)b * nq + qi;
  bf16x8 qa[2];
  load_half_row_bf<Q16>(q, qrow, ldq, h, half, qa);
  // ---- stage K, V of head (b,h): rows beyond nk are zero (their scores are masked out below)
  for (int c = threadIdx.x; c < nkp * (D / 4); c += 256) {
    const int key = c / (D / 4), d4 = (c - key * (D / 4)) * 4;
    short4 kb = make_short4(0, 0, 0, 0), vb = kb;
    if (key < nk) {
      const long long krow = (long long)b * nk + key;
      kb = load4_bf<KV16>(k, krow * ldk + h * D + d4);
      vb = load4_bf<KV16>(v, krow * ldv + h * D + d4);
    }
    *reinterpret_cast<short4 *>(sK + key * KS + d4) = kb;
    sV[(d4 + 0) * VS + key] = vb.x;
    sV[(d4 + 1) * VS + key] = vb.y;
    sV[(d4 + 2) * VS + key] = vb.z;
    sV[(d4 + 3) * VS + key] = vb.w;
  }
  __syncthreads();
  if (q0 >= nq) return;  // whole wave (no barrier after this point)

  f32x16 o = zero16();
  float m = -__builtin_inff(), l = 0.f;
  const long long bias_row = (((long long)b * H + h) * nq + qi) * nk;
  for (int k0 = 0; k0 < nk; k0 += 32) {
    f32x16 s = zero16();
    float bt[16];
    load_bias_tile(bias, bias_mode, bias_row, k0, half, nk, bt);
    const short *kr = sK + (k0 + r) * KS + 16 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(kr + 8 * t);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qa[t], s, 0, 0, 0);  // s[reg] = S[query r][key k0 + acc_row]
    }
    float tmax = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, half);
      float x = -__builtin_inff();
      if (key < nk) {
        x = apply_bias_value(s[i] * scale, bias_mode, bt[i]);
        if (mask != nullptr && mask[(long long)b * nk + key] == 0.f) x = -10000.f;
      }
      s[i] = x;
      tmax = fmaxf(tmax, x);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    const float alpha = __expf(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __expf(s[i] - m_new);
      s[i] = p;
      psum += p;
      o[i] *= alpha;
    }
    l = l * alpha + psum;
    m = m_new;
    // O^T[dim][query] += V^T[dim][key] * P^T[key][query]: K-step t sums over keys k0 + acc_row(8t + j, half), i.e. the
    // two 4-key groups k0 + 16t + 4*half + {0..3} and + 8 + {0..3} of V^T row `r`
    const short *vr = sV + r * VS + k0 + 4 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 va, pb;
      const short4 g0 = *reinterpret_cast<const short4 *>(vr + 16 * t);
      const short4 g1 = *reinterpret_cast<const short4 *>(vr + 16 * t + 8);
      va[0] = g0.x; va[1] = g0.y; va[2] = g0.z; va[3] = g0.w;
      va[4] = g1.x; va[5] = g1.y; va[6] = g1.z; va[7] = g1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = bf16_bits(s[8 * t + j]);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o, 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 32);
  if (q0 + r < nq) {
    const float inv = 1.f / l;
    if (O16) {
      short *__restrict__ orow = reinterpret_cast<short *>(out) + qrow * HD + h * D;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        short4 w;
        w.x = bf16_bits(o[4 * g + 0] * inv); w.y = bf16_bits(o[4 * g + 1] * inv);
        w.z = bf16_bits(o[4 * g + 2] * inv); w.w = bf16_bits(o[4 * g + 3] * inv);
        *reinterpret_cast<short4 *>(orow + 8 * g + 4 * half) = w;
      }
    } else {
      float *__restrict__ orow = reinterpret_cast<float *>(out) + qrow * HD + h * D;
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 = dims 8g + 4*half + 0..3
        float4 w;
        w.x = o[4 * g + 0] * inv; w.y = o[4 * g + 1] * inv; w.z = o[4 * g + 2] * inv; w.w = o[4 * g + 3] * inv;
        *reinterpret_cast<float4 *>(orow + 8 * g + 4 * half) = w;
      }
    }
    if (half == 0) lse[((long long)b * H + h) * nq + q0 + r] = m + __logf(l);
  }
}

// ---------------------------------------------------------------------------------------------
// backward 1: wave = 32 queries; writes dQ, delta = rowsum(dO*O) and (optionally) dbias.
// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(64) void sdpa_bwd_dq_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ bias, int bias_mode, const float *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ lse, const float *__restrict__ dout, int H, int nq, int nk, int ldq, int ldk, int ldv,
    float scale, float *__restrict__ dq, float *__restrict__ dbias, float *__restrict__ delta) {
  const int lane = threadIdx.x, r = lane & 31, half = lane >> 5;
  const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int HD = H * D;
  const int qi = min(q0 + r, nq - 1);
  const long long qrow = (long long)b * nq + qi;
  const bool q_ok = q0 + r < nq;

  float qreg[16], doreg[16], oreg[16];
  load_half_row(q, qrow, ldq, h, half, qreg);
  load_half_row(dout, qrow, HD, h, half, doreg);
  load_half_row(out, qrow, HD, h, half, oreg);
  float dl = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dl += doreg[i] * oreg[i];
  dl += __shfl_xor(dl, 32);
  const long long stat = ((long long)b * H + h) * nq + qi;
  if (q_ok && half == 0) delta[stat] = dl;
  const float lse_q = lse[stat];
  const long long bias_row = stat * nk;

  f32x16 dqa = zero16();
  float kreg[16], vreg[16], kcol[16];
  auto load_tile = [&](int k0, float (&kr)[16], float (&vr)[16], float (&kc)[16]) {
    const long long krow = (long long)b * nk + min(k0 + r, nk - 1);
    load_half_row(k, krow, ldk, h, half, kr);
    load_half_row(v, krow, ldv, h, half, vr);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = min(k0 + acc_row(i, half), nk - 1);
      kc[i] = k[((long long)b * nk + key) * ldk + h * D + r];
    }
  };
  load_tile(0, kreg, vreg, kcol);
  float bnext[16];  // the next tile's bias values travel with its K / V rows (16 B / lane x 4 loads behind two products otherwise)
#pragma unroll
  for (int i = 0; i < 16; ++i) bnext[i] = 0.f;
  load_bias_tile(bias, bias_mode, bias_row, 0, half, nk, bnext);
  for (int k0 = 0; k0 < nk; k0 += 32) {
    float knext[16], vnext[16], kcnext[16];  // next tile in flight while this one is computed
    load_tile(min(k0 + 32, nk - 1), knext, vnext, kcnext);
    float bt[16], db[16] = {};
#pragma unroll
    for (int i = 0; i < 16; ++i) bt[i] = bnext[i];
    if (k0 + 32 < nk) load_bias_tile(bias, bias_mode, bias_row, k0 + 32, half, nk, bnext);
    f32x16 s = mfma_rows<BF>(kreg, qreg, zero16());    // S^T  [key][query]
    f32x16 dp = mfma_rows<BF>(vreg, doreg, zero16());  // dP^T [key][query] = V dO^T
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, half);
      float ds = 0.f;
      if (key < nk) {
        const float raw = s[i] * scale;
        float x = apply_bias_value(raw, bias_mode, bt[i]);
        const bool masked = mask != nullptr && mask[(long long)b * nk + key] == 0.f;
        if (masked) x = -10000.f;
        const float p = __expf(x - lse_q);
        ds = masked ? 0.f : p * (dp[i] - dl);  // d/d(pre-softmax score); masked_fill blocks the gradient
        db[i] = bias_mode == 2 ? ds * raw : ds;
        if (bias_mode == 2) ds *= bt[i];
      }
      s[i] = ds;
    }
    if (dbias != nullptr && q_ok) store_dbias_tile(dbias, bias_row, k0, half, nk, db);
    // dQ^T[dim][query] += K^T[dim][key] * dS^T[key][query]
    dqa = mfma_rows<BF>(kcol, s, dqa);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      kreg[i] = knext[i];
      vreg[i] = vnext[i];
      kcol[i] = kcnext[i];
    }
  }
  if (q_ok) {
    float *__restrict__ row = dq + qrow * ldq + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w;
      w.x = dqa[4 * g + 0] * scale; w.y = dqa[4 * g + 1] * scale; w.z = dqa[4 * g + 2] * scale;
      w.w = dqa[4 * g + 3] * scale;
      *reinterpret_cast<float4 *>(row + 8 * g + 4 * half) = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 1, bf16-MFMA form with the head's K and V shared through LDS (same staging idea as sdpa_fwd_lds_kernel):
// K row-major (S^T = K Q^T), V row-major (dP^T = V dO^T) and K transposed (dQ^T += K^T dS^T), all bf16, converted once.
// ---------------------------------------------------------------------------------------------
template <int IO>  // as sdpa_fwd_lds_kernel: bit 0 q, bit 1 k / v, bit 2 out hold bf16 (dout and the gradients stay fp32)
__global__ __launch_bounds__(256) void sdpa_bwd_dq_lds_kernel(
    const void *__restrict__ q, const void *__restrict__ k, const void *__restrict__ v,
    const float *__restrict__ bias, int bias_mode, const float *__restrict__ mask, const void *__restrict__ out,
    const float *__restrict__ lse, const float *__restrict__ dout, int H, int nq, int nk, int nkp, int ldq, int ldk,
    int ldv, float scale, float *__restrict__ dq, float *__restrict__ dbias, float *__restrict__ delta) {
  extern __shared__ __attribute__((aligned(16))) short sm_kv[];
  constexpr int KS = D + 8;
  const int TS = nkp + 4;
  short *sK = sm_kv;                 // [nkp][KS]
  short *sV = sK + nkp * KS;         // [nkp][KS]
  short *sKt = sV + nkp * KS;        // [D][TS]
  constexpr bool Q16 = (IO & 1) != 0, KV16 = (IO & 2) != 0, O16 = (IO & 4) != 0;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, h, b;
  xcd_block(bx, h, b);
  const int HD = H * D;
  for (int c = threadIdx.x; c < nkp * (D / 4); c += 256) {
    const int key = c / (D / 4), d4 = (c - key * (D / 4)) * 4;
    short4 kb = make_short4(0, 0, 0, 0), vb = kb;
    if (key < nk) {
      const long long krow = (long long)b * nk + key;
      kb = load4_bf<KV16>(k, krow * ldk + h * D + d4);
      vb = load4_bf<KV16>(v, krow * ldv + h * D + d4);
    }
    *reinterpret_cast<short4 *>(sK + key * KS + d4) = kb;
    *reinterpret_cast<short4 *>(sV + key * KS + d4) = vb;
    sKt[(d4 + 0) * TS + key] = kb.x;
    sKt[(d4 + 1) * TS + key] = kb.y;
    sKt[(d4 + 2) * TS + key] = kb.z;
    sKt[(d4 + 3) * TS + key] = kb.w;
  }
  __syncthreads();
  const int q0 = (bx * 4 + wave) * 32;
  if (q0 >= nq) return;
  const int qi = min(q0 + r, nq - 1);
  const long long qrow = (long long)b * nq + qi;
  const bool q_ok = q0 + r < nq;
  float doreg[16], oreg[16];
  bf16x8 qa[2], da[2];
  load_half_row_bf<Q16>(q, qrow, ldq, h, half, qa);
  load_half_row(dout, qrow, HD, h, half, doreg);
  load_half_row_f<O16>(out, qrow, HD, h, half, oreg);
  float dl = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dl += doreg[i] * oreg[i];
  dl += __shfl_xor(dl, 32);
  const long long stat = ((long long)b * H + h) * nq + qi;
  if (q_ok && half == 0) delta[stat] = dl;
  const float lse_q = lse[stat];
  const long long bias_row = stat * nk;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) da[t][j] = bf16_bits(doreg[8 * t + j]);
  f32x16 dqa = zero16();
  for (int k0 = 0; k0 < nk; k0 += 32) {
    f32x16 s = zero16(), dp = zero16();
    float bt[16], db[16] = {};
    load_bias_tile(bias, bias_mode, bias_row, k0, half, nk, bt);
    const short *kr = sK + (k0 + r) * KS + 16 * half, *vr = sV + (k0 + r) * KS + 16 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(kr + 8 * t), qa[t], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(vr + 8 * t), da[t], dp, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, half);
      float ds = 0.f;
      if (key < nk) {
        const float raw = s[i] * scale;
        float x = apply_bias_value(raw, bias_mode, bt[i]);
        const bool masked = mask != nullptr && mask[(long long)b * nk + key] == 0.f;
        if (masked) x = -10000.f;
        const float p = __expf(x - lse_q);
        ds = masked ? 0.f : p * (dp[i] - dl);
        db[i] = bias_mode == 2 ? ds * raw : ds;
        if (bias_mode == 2) ds *= bt[i];
      }
      s[i] = ds;
    }
    if (dbias != nullptr && q_ok) store_dbias_tile(dbias, bias_row, k0, half, nk, db);
    // dQ^T[dim][query] += K^T[dim][key] * dS^T[key][query], keys in the accumulator's row order (see the forward)
    const short *kt = sKt + r * TS + k0 + 4 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 ka, sb;
      const short4 g0 = *reinterpret_cast<const short4 *>(kt + 16 * t);
      const short4 g1 = *reinterpret_cast<const short4 *>(kt + 16 * t + 8);
      ka[0] = g0.x; ka[1] = g0.y; ka[2] = g0.z; ka[3] = g0.w;
      ka[4] = g1.x; ka[5] = g1.y; ka[6] = g1.z; ka[7] = g1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) sb[j] = bf16_bits(s[8 * t + j]);
      dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, sb, dqa, 0, 0, 0);
    }
  }
  if (q_ok) {
    float *__restrict__ row = dq + qrow * ldq + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w;
      w.x = dqa[4 * g + 0] * scale; w.y = dqa[4 * g + 1] * scale; w.z = dqa[4 * g + 2] * scale;
      w.w = dqa[4 * g + 3] * scale;
      *reinterpret_cast<float4 *>(row + 8 * g + 4 * half) = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 2: wave = 32 keys; loops over query tiles; writes dK, dV (no atomics).
// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void sdpa_bwd_dkv_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ bias, int bias_mode, const float *__restrict__ mask, const float *__restrict__ lse,
    const float *__restrict__ dout, const float *__restrict__ delta, int H, int nq, int nk, int ldq, int ldk, int ldv,
    float scale, float *__restrict__ dk, float *__restrict__ dv) {
  const int lane = threadIdx.x, r = lane & 31, half = lane >> 5;
  const int k0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int HD = H * D;
  const int ki = min(k0 + r, nk - 1);
  const long long krow = (long long)b * nk + ki;
  const bool k_ok = k0 + r < nk;
  const bool masked = mask != nullptr && mask[(long long)b * nk + ki] == 0.f;

  float kreg[16], vreg[16];
  load_half_row(k, krow, ldk, h, half, kreg);
  load_half_row(v, krow, ldv, h, half, vreg);

  f32x16 dka = zero16(), dva = zero16();
  // Rows of the next query tile are prefetched one tile ahead; the column-layout copies and the per-query
  // statistics of the CURRENT tile are requested at the top of the iteration and consumed after the two S / dP
  // products (double-buffering them as well costs 64 more VGPRs and drops the kernel to one wave per SIMD).
  float qreg[16], doreg[16];
  auto load_rows = [&](int q0, float (&qr)[16], float (&dr)[16]) {
    const long long qrow = (long long)b * nq + min(q0 + r, nq - 1);
    load_half_row(q, qrow, ldq, h, half, qr);
    load_half_row(dout, qrow, HD, h, half, dr);
  };
  load_rows(0, qreg, doreg);
  for (int q0 = 0; q0 < nq; q0 += 32) {
    float qn[16], dn[16], docol[16], qcol[16], lse_t[16], delta_t[16];
    load_rows(min(q0 + 32, nq - 1), qn, dn);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = min(q0 + acc_row(i, half), nq - 1);
      const long long qr_ = (long long)b * nq + qq;
      docol[i] = dout[qr_ * HD + h * D + r];
      qcol[i] = q[qr_ * ldq + h * D + r];
      const long long stat = ((long long)b * H + h) * nq + qq;
      lse_t[i] = lse[stat];
      delta_t[i] = delta[stat];
    }
    f32x16 s = mfma_rows<BF>(qreg, kreg, zero16());    // S  [query][key], lane = key
    f32x16 dp = mfma_rows<BF>(doreg, vreg, zero16());  // dP [query][key]
    f32x16 p;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = q0 + acc_row(i, half);
      float pv = 0.f, ds = 0.f;
      if (qq < nq && k_ok) {
        const long long stat = ((long long)b * H + h) * nq + qq;
        const float raw = s[i] * scale;
        float x = apply_bias(raw, bias_mode, bias, stat * nk + ki);
        if (masked) x = -10000.f;
        pv = __expf(x - lse_t[i]);
        ds = masked ? 0.f : pv * (dp[i] - delta_t[i]);
        if (bias_mode == 2) ds *= bias[stat * nk + ki];
      }
      p[i] = pv;
      s[i] = ds;
    }
    // dV^T[dim][key] += dO^T[dim][query] * P[query][key];  dK^T[dim][key] += Q^T[dim][query] * dS[query][key]
    dva = mfma_rows<BF>(docol, p, dva);
    dka = mfma_rows<BF>(qcol, s, dka);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      qreg[i] = qn[i];
      doreg[i] = dn[i];
    }
  }
  if (k_ok) {
    float *__restrict__ rk = dk + krow * ldk + h * D;
    float *__restrict__ rv = dv + krow * ldv + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a, c;
      a.x = dka[4 * g + 0] * scale; a.y = dka[4 * g + 1] * scale; a.z = dka[4 * g + 2] * scale;
      a.w = dka[4 * g + 3] * scale;
      c.x = dva[4 * g + 0]; c.y = dva[4 * g + 1]; c.z = dva[4 * g + 2]; c.w = dva[4 * g + 3];
      *reinterpret_cast<float4 *>(rk + 8 * g + 4 * half) = a;
      *reinterpret_cast<float4 *>(rv + 8 * g + 4 * half) = c;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 2, bf16-MFMA form with the head's Q and dO shared through LDS: every wave of the one-wave form streamed all
// queries of its head from L2 (64 KB per wave, eight key-tile waves per head re-reading the same rows, plus 32 strided
// dword loads per tile for the column-layout copies).  A workgroup = up to four key tiles of one (b,h); Q and dO are
// converted to bf16 ONCE while staging, row-major (S = Q K^T, dP = dO V^T: a lane's operand = two ds_read_b128) and
// transposed (dV^T += dO^T P, dK^T += Q^T dS: four ds_read_b64 in the accumulator's row order); lse / delta as fp32.
// ---------------------------------------------------------------------------------------------
template <bool BIAS, int IO = 0>  // BIAS = false: no bias registers in the match cores' instantiation (16 more cost them 12 %)
__global__ __launch_bounds__(512) void sdpa_bwd_dkv_lds_kernel(
    const void *__restrict__ q, const void *__restrict__ k, const void *__restrict__ v,
    const float *__restrict__ bias, int bias_mode, const float *__restrict__ mask, const float *__restrict__ lse,
    const float *__restrict__ dout, const float *__restrict__ delta, int H, int nq, int nqp, int nk, int ldq, int ldk,
    int ldv, float scale, float *__restrict__ dk, float *__restrict__ dv, int wpb, int qsplit) {
  extern __shared__ __attribute__((aligned(16))) short sm_q[];
  constexpr int KS = D + 8;
  const int TS = nqp + 4;
  short *sQ = sm_q;                  // [nqp][KS]
  short *sDO = sQ + nqp * KS;        // [nqp][KS]
  short *sQt = sDO + nqp * KS;       // [D][TS]
  short *sDOt = sQt + D * TS;        // [D][TS]
  float *sL = reinterpret_cast<float *>(sDOt + D * TS);  // [nqp] lse, then [nqp] delta (offset is a multiple of 16 bytes)
  float *sDl = sL + nqp;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, h, b;
  xcd_block(bx, h, b);
  const int HD = H * D, nthr = blockDim.x;
  for (int c = threadIdx.x; c < nqp * (D / 4); c += nthr) {
    const int qi = c / (D / 4), d4 = (c - qi * (D / 4)) * 4;
    float4 dv4 = make_float4(0.f, 0.f, 0.f, 0.f);
    short4 qb = make_short4(0, 0, 0, 0), db;
    if (qi < nq) {
      const long long qrow = (long long)b * nq + qi;
      qb = load4_bf<(IO & 1) != 0>(q, qrow * ldq + h * D + d4);
      dv4 = *reinterpret_cast<const float4 *>(dout + qrow * HD + h * D + d4);
    }
    db.x = bf16_bits(dv4.x); db.y = bf16_bits(dv4.y); db.z = bf16_bits(dv4.z); db.w = bf16_bits(dv4.w);
    *reinterpret_cast<short4 *>(sQ + qi * KS + d4) = qb;
    *reinterpret_cast<short4 *>(sDO + qi * KS + d4) = db;
    sQt[(d4 + 0) * TS + qi] = qb.x; sQt[(d4 + 1) * TS + qi] = qb.y; sQt[(d4 + 2) * TS + qi] = qb.z; sQt[(d4 + 3) * TS + qi] = qb.w;
    sDOt[(d4 + 0) * TS + qi] = db.x; sDOt[(d4 + 1) * TS + qi] = db.y; sDOt[(d4 + 2) * TS + qi] = db.z; sDOt[(d4 + 3) * TS + qi] = db.w;
  }
  for (int c = threadIdx.x; c < nqp; c += nthr) {
    const long long stat = ((long long)b * H + h) * nq + min(c, nq - 1);
    sL[c] = lse[stat];
    sDl[c] = delta[stat];
  }
  __syncthreads();
  // wave -> (key tile kt of the workgroup, query split qs): the qsplit waves of a key tile walk interleaved query tiles and
  // their partial dK / dV meet in LDS at the end — with one wave per key tile a (batch, head) pair of the relation module
  // (8 key tiles x 8 dependent query iterations) kept 64 workgroups of 4 waves busy, a quarter of the chip at one wave per
  // SIMD.  No wave leaves before the last barrier.
  const int kt = wave % wpb, qs = wave / wpb;
  const int k0 = (bx * wpb + kt) * 32;
  const bool tile_ok = k0 < nk;
  const int ki = min(k0 + r, nk - 1);
  const long long krow = (long long)b * nk + ki;
  const bool k_ok = k0 + r < nk;
  const bool masked = mask != nullptr && mask[(long long)b * nk + ki] == 0.f;
  bf16x8 kb[2], vb[2];
  load_half_row_bf<(IO & 2) != 0>(k, krow, ldk, h, half, kb);
  load_half_row_bf<(IO & 2) != 0>(v, krow, ldv, h, half, vb);
  const long long stat0 = ((long long)b * H + h) * nq;
  f32x16 dka = zero16(), dva = zero16();
  for (int q0 = tile_ok ? 32 * qs : nq; q0 < nq; q0 += 32 * qsplit) {
    f32x16 s = zero16(), dp = zero16();
    // the tile's 16 bias values of this lane (key ki, queries q0 + acc_row): ALL loads issued here, in front of the products,
    // under one wave-uniform branch (read where they are used, each sat behind its own bounds test: 16 dependent L2 round
    // trips per tile — the relation module's cores took 40 us for 0.17 GF)
    float bt[BIAS ? 16 : 1];
    if (BIAS) {
#pragma unroll
      for (int i = 0; i < 16; ++i) bt[i] = bias[(stat0 + min(q0 + acc_row(i, half), nq - 1)) * nk + ki];
    }
    const short *qr = sQ + (q0 + r) * KS + 16 * half, *dr = sDO + (q0 + r) * KS + 16 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // S[query][key], dP[query][key]: lane = key
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(qr + 8 * t), kb[t], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(dr + 8 * t), vb[t], dp, 0, 0, 0);
    }
    f32x16 p;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4 *>(sL + q0 + 8 * g + 4 * half);
      const float4 d4 = *reinterpret_cast<const float4 *>(sDl + q0 + 8 * g + 4 * half);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e;
        const int qq = q0 + acc_row(i, half);
        float pv = 0.f, ds = 0.f;
        if (qq < nq && k_ok) {
          const float raw = s[i] * scale;
          float x = BIAS ? apply_bias_value(raw, bias_mode, bt[BIAS ? i : 0]) : raw;
          if (masked) x = -10000.f;
          pv = __expf(x - lv[e]);
          ds = masked ? 0.f : pv * (dp[i] - dl[e]);
          if (BIAS && bias_mode == 2) ds *= bt[BIAS ? i : 0];
        }
        p[i] = pv;
        s[i] = ds;
      }
    }
    // dV^T[dim][key] += dO^T[dim][query] P[query][key];  dK^T[dim][key] += Q^T[dim][query] dS[query][key]
    const short *dot = sDOt + r * TS + q0 + 4 * half, *qt = sQt + r * TS + q0 + 4 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 da, qa, pb, sb;
      const short4 a0 = *reinterpret_cast<const short4 *>(dot + 16 * t), a1 = *reinterpret_cast<const short4 *>(dot + 16 * t + 8);
      const short4 c0 = *reinterpret_cast<const short4 *>(qt + 16 * t), c1 = *reinterpret_cast<const short4 *>(qt + 16 * t + 8);
      da[0] = a0.x; da[1] = a0.y; da[2] = a0.z; da[3] = a0.w; da[4] = a1.x; da[5] = a1.y; da[6] = a1.z; da[7] = a1.w;
      qa[0] = c0.x; qa[1] = c0.y; qa[2] = c0.z; qa[3] = c0.w; qa[4] = c1.x; qa[5] = c1.y; qa[6] = c1.z; qa[7] = c1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pb[j] = bf16_bits(p[8 * t + j]);
        sb[j] = bf16_bits(s[8 * t + j]);
      }
      dva = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, pb, dva, 0, 0, 0);
      dka = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, sb, dka, 0, 0, 0);
    }
  }
  // tree over the query splits through LDS (the staged operands are dead after the barrier): slot = 32 floats per lane
  float *red = reinterpret_cast<float *>(sm_q);
  for (int step = qsplit >> 1; step >= 1; step >>= 1) {
    __syncthreads();
    if (qs >= step && qs < 2 * step) {
      float *slot = red + (size_t)((qs - step) * wpb + kt) * 2048 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        slot[64 * i] = dka[i];
        slot[64 * (16 + i)] = dva[i];
      }
    }
    __syncthreads();
    if (qs < step) {
      const float *slot = red + (size_t)(qs * wpb + kt) * 2048 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        dka[i] += slot[64 * i];
        dva[i] += slot[64 * (16 + i)];
      }
    }
  }
  if (k_ok && qs == 0) {
    float *__restrict__ rk = dk + krow * ldk + h * D;
    float *__restrict__ rv = dv + krow * ldv + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a, c;
      a.x = dka[4 * g + 0] * scale; a.y = dka[4 * g + 1] * scale; a.z = dka[4 * g + 2] * scale;
      a.w = dka[4 * g + 3] * scale;
      c.x = dva[4 * g + 0]; c.y = dva[4 * g + 1]; c.z = dva[4 * g + 2]; c.w = dva[4 * g + 3];
      *reinterpret_cast<float4 *>(rk + 8 * g + 4 * half) = a;
      *reinterpret_cast<float4 *>(rv + 8 * g + 4 * half) = c;
    }
  }
}

// row strides (floats) of q / k / v (and of dq / dk / dv): >= H*D and a multiple of 4 (16-byte row slices)
int sdpa_lds_min_bh() {
  static const int v = getenv("VLP3D_SDPA_LDS_MIN_BH") ? atoi(getenv("VLP3D_SDPA_LDS_MIN_BH")) : 64;
  return v;
}

// Workgroup shape of the LDS dK/dV kernel: key tiles per workgroup x query splits per key tile.  Measured (dq + dkv, us) at
// (B*H, nq, nk) = (256, 256, 256) / (256, 256, 49) / (32, 256, 256 + bias): 4x1 61.0 / 42.4 / 95.7, 4x2 56.0 / 32.0 / 73.5,
// 2x4 64.7 / 26.0 / 61.8, 1x8 82.0 / 25.8 / 56.1 — many key-tile waves: share the staging; few: split the queries.
int dkv_wpb(long long key_tile_waves) {
  static const int v = getenv("VLP3D_SDPA_DKV_WPB") ? atoi(getenv("VLP3D_SDPA_DKV_WPB")) : 0;
  if (v > 0) return v > 4 ? 4 : v;
  return 2;  // (round-4 sweep inside the step: 2 key tiles x up to 4 query splits for every shape of the path; round 3: 4 / 1)
}
int dkv_qsplit(long long key_tile_waves) {  // power of two
  static const int v = getenv("VLP3D_SDPA_DKV_QSPLIT") ? atoi(getenv("VLP3D_SDPA_DKV_QSPLIT")) : 0;
  if (v > 0) return v >= 8 ? 8 : (v >= 4 ? 4 : (v >= 2 ? 2 : 1));
  return key_tile_waves >= 2048 ? 2 : 8;
}

bool bad_ld(int ldq, int ldk, int ldv, int H) {
  return ldq < H * D || ldk < H * D || ldv < H * D || ((ldq | ldk | ldv) & 3);
}

bool bad(int B, int H, int nq, int nk, int Dh, int bias_mode) {
  return B < 1 || B > 65535 || H < 1 || H > 65535 || nq < 1 || nk < 1 || Dh != D || bias_mode < 0 || bias_mode > 2;
}

}  // namespace

// io bits (bf16-MFMA LDS kernels only): 1 = q, 2 = k and v, 4 = out hold bf16 rows instead of fp32 (row strides in
// elements; a bf16 operand needs 16-byte rows: stride % 8 == 0).  Combinations built: 0, 5 (cross-attention: q and out of the
// 16 384-row side, k / v from the small fp32 projection of the tokens), 7 (self-attention on a merged bf16 q|k|v).
static bool io_ok(int io) { return io == 0 || io == 5 || io == 7; }
static bool bad_ld_io(int ldq, int ldk, int ldv, int io) {
  return ((io & 1) && (ldq & 7)) || ((io & 2) && ((ldk | ldv) & 7));
}

extern "C" int vlp3d_sdpa_fwd_io(const void *q, const void *k, const void *v, const float *bias, int bias_mode,
                                 const float *mask, int B, int H, int nq, int nk, int Dh, void *out, float *lse,
                                 int bf16_mma, int ldq, int ldk, int ldv, int io, void *stream) {
  if (!q || !k || !v || !out || !lse || bad(B, H, nq, nk, Dh, bias_mode) || (bias_mode != 0 && !bias))
    return VLP3D_EINVAL;
  if (bad_ld(ldq, ldk, ldv, H) || !io_ok(io) || bad_ld_io(ldq, ldk, ldv, io)) return VLP3D_EINVAL;
  if (io != 0 && !(bf16_mma && nk <= 384)) return VLP3D_EINVAL;
  const float scale = 1.0f / sqrtf((float)Dh);
  const dim3 grid(vlp3d_cdiv(nq, 32), H, B);
  const float *qf = reinterpret_cast<const float *>(q), *kf = reinterpret_cast<const float *>(k),
              *vf = reinterpret_cast<const float *>(v);
  float *of = reinterpret_cast<float *>(out);
  if (bf16_mma && nk <= 384) {  // K/V of a head shared by 4 waves through LDS (<= 55 KB)
    const int nkp = (nk + 31) & ~31;
    const size_t lds = ((size_t)nkp * (D + 8) + (size_t)D * (nkp + 4)) * sizeof(short);
    const dim3 g4(vlp3d_cdiv(nq, 128), H, B);
#define VLP3D_SDPA_F(IOv) hipLaunchKernelGGL(sdpa_fwd_lds_kernel<IOv>, g4, dim3(256), lds, (hipStream_t)stream, q, k, v, \
                                             bias, bias_mode, mask, H, nq, nk, nkp, ldq, ldk, ldv, scale, out, lse)
    if (io == 5) VLP3D_SDPA_F(5);
    else if (io == 7) VLP3D_SDPA_F(7);
    else VLP3D_SDPA_F(0);
#undef VLP3D_SDPA_F
  } else if (bf16_mma)
    hipLaunchKernelGGL(sdpa_fwd_kernel<true>, grid, dim3(64), 0, (hipStream_t)stream, qf, kf, vf, bias, bias_mode, mask, H,
                       nq, nk, ldq, ldk, ldv, scale, of, lse);
  else
    hipLaunchKernelGGL(sdpa_fwd_kernel<false>, grid, dim3(64), 0, (hipStream_t)stream, qf, kf, vf, bias, bias_mode, mask,
                       H, nq, nk, ldq, ldk, ldv, scale, of, lse);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_sdpa_fwd(const float *q, const float *k, const float *v, const float *bias, int bias_mode,
                              const float *mask, int B, int H, int nq, int nk, int Dh, float *out, float *lse,
                              int bf16_mma, int ldq, int ldk, int ldv, void *stream) {
  return vlp3d_sdpa_fwd_io(q, k, v, bias, bias_mode, mask, B, H, nq, nk, Dh, out, lse, bf16_mma, ldq, ldk, ldv, 0, stream);
}

extern "C" int vlp3d_sdpa_bwd_io(const void *q, const void *k, const void *v, const float *bias, int bias_mode,
                                 const float *mask, const void *out, const float *lse, const float *dout, int B, int H,
                                 int nq, int nk, int Dh, float *dq, float *dk, float *dv, float *dbias, float *delta,
                                 int bf16_mma, int ldq, int ldk, int ldv, int io, void *stream) {
  if (!q || !k || !v || !out || !lse || !dout || !dq || !dk || !dv || !delta || bad(B, H, nq, nk, Dh, bias_mode) ||
      (bias_mode != 0 && !bias))
    return VLP3D_EINVAL;
  if (bad_ld(ldq, ldk, ldv, H) || !io_ok(io) || bad_ld_io(ldq, ldk, ldv, io)) return VLP3D_EINVAL;
  const float scale = 1.0f / sqrtf((float)Dh);
  hipStream_t s = (hipStream_t)stream;
  const dim3 gq(vlp3d_cdiv(nq, 32), H, B), gk(vlp3d_cdiv(nk, 32), H, B);
  const float *qf = reinterpret_cast<const float *>(q), *kf = reinterpret_cast<const float *>(k),
              *vf = reinterpret_cast<const float *>(v), *of = reinterpret_cast<const float *>(out);
  const int nqp = (nq + 31) & ~31;
  const size_t lds_kv = ((size_t)2 * nqp * (D + 8) + (size_t)2 * D * (nqp + 4)) * sizeof(short) + (size_t)2 * nqp * sizeof(float);
  const bool dq_lds = nk <= 288 && (long long)B * H >= sdpa_lds_min_bh();
  const bool dkv_lds = nq <= 512 && lds_kv <= 150 * 1024;
  if (io != 0 && !(bf16_mma && dq_lds && dkv_lds && bias_mode == 0)) return VLP3D_EINVAL;  // bf16 rows: the LDS kernels only
  if (bf16_mma) {
    if (dq_lds) {  // K, V and K^T of a head staged once in LDS (<= 63 KB) and shared by
      // four query waves; with few (batch, head) pairs the one-wave form keeps more CUs busy
      const int nkp = (nk + 31) & ~31;
      const size_t lds = ((size_t)2 * nkp * (D + 8) + (size_t)D * (nkp + 4)) * sizeof(short);
      const dim3 g4(vlp3d_cdiv(nq, 128), H, B);
#define VLP3D_SDPA_DQ(IOv) hipLaunchKernelGGL(sdpa_bwd_dq_lds_kernel<IOv>, g4, dim3(256), lds, s, q, k, v, bias, \
                                              bias_mode, mask, out, lse, dout, H, nq, nk, nkp, ldq, ldk, ldv, scale, dq, dbias, delta)
      if (io == 5) VLP3D_SDPA_DQ(5);
      else if (io == 7) VLP3D_SDPA_DQ(7);
      else VLP3D_SDPA_DQ(0);
#undef VLP3D_SDPA_DQ
    } else
      hipLaunchKernelGGL(sdpa_bwd_dq_kernel<true>, gq, dim3(64), 0, s, qf, kf, vf, bias, bias_mode, mask, of, lse, dout, H,
                         nq, nk, ldq, ldk, ldv, scale, dq, dbias, delta);
    if (dkv_lds) {  // Q, dO (+ transposes) of a head staged once per workgroup of up to 4 key tiles
      // workgroup = wpb key tiles x qsplit query splits, at most 8 waves (174 registers per lane)
      const int tiles = vlp3d_cdiv(nk, 32), qtiles = nqp / 32;
      const long long ktw = (long long)B * H * tiles;
      int wpb = dkv_wpb(ktw), qsplit = dkv_qsplit(ktw);
      if (wpb > tiles) wpb = tiles;
      while (qsplit > qtiles) qsplit >>= 1;
      while (wpb * qsplit > 8) {
        if (qsplit > 1 && qsplit >= wpb) qsplit >>= 1;
        else --wpb;
      }
      size_t lds_all = lds_kv;
      const size_t lds_red = (size_t)(qsplit >> 1) * wpb * 2048 * sizeof(float);
      if (lds_red > lds_all) lds_all = lds_red;
      const bool with_bias = bias_mode != 0;
      const void *fn = with_bias ? reinterpret_cast<const void *>(sdpa_bwd_dkv_lds_kernel<true, 0>)
                       : io == 5 ? reinterpret_cast<const void *>(sdpa_bwd_dkv_lds_kernel<false, 5>)
                       : io == 7 ? reinterpret_cast<const void *>(sdpa_bwd_dkv_lds_kernel<false, 7>)
                                 : reinterpret_cast<const void *>(sdpa_bwd_dkv_lds_kernel<false, 0>);
      if (lds_all > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all);
        if (e != hipSuccess) return (int)e;
      }
      const dim3 gd(vlp3d_cdiv(tiles, wpb), H, B), bd(64 * wpb * qsplit);
#define VLP3D_SDPA_DKV(BIASv, IOv) hipLaunchKernelGGL((sdpa_bwd_dkv_lds_kernel<BIASv, IOv>), gd, bd, lds_all, s, q, k, v, bias, \
                                                      bias_mode, mask, lse, dout, delta, H, nq, nqp, nk, ldq, ldk, ldv, scale, dk, dv, wpb, qsplit)
      if (with_bias) VLP3D_SDPA_DKV(true, 0);
      else if (io == 5) VLP3D_SDPA_DKV(false, 5);
      else if (io == 7) VLP3D_SDPA_DKV(false, 7);
      else VLP3D_SDPA_DKV(false, 0);
#undef VLP3D_SDPA_DKV
    } else
      hipLaunchKernelGGL(sdpa_bwd_dkv_kernel<true>, gk, dim3(64), 0, s, qf, kf, vf, bias, bias_mode, mask, lse, dout, delta,
                         H, nq, nk, ldq, ldk, ldv, scale, dk, dv);
  } else {
    hipLaunchKernelGGL(sdpa_bwd_dq_kernel<false>, gq, dim3(64), 0, s, qf, kf, vf, bias, bias_mode, mask, of, lse, dout, H,
                       nq, nk, ldq, ldk, ldv, scale, dq, dbias, delta);
    hipLaunchKernelGGL(sdpa_bwd_dkv_kernel<false>, gk, dim3(64), 0, s, qf, kf, vf, bias, bias_mode, mask, lse, dout, delta,
                       H, nq, nk, ldq, ldk, ldv, scale, dk, dv);
  }
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_sdpa_bwd(const float *q, const float *k, const float *v, const float *bias, int bias_mode,
                              const float *mask, const float *out, const float *lse, const float *dout, int B, int H,
                              int nq, int nk, int Dh, float *dq, float *dk, float *dv, float *dbias, float *delta,
                              int bf16_mma, int ldq, int ldk, int ldv, void *stream) {
  return vlp3d_sdpa_bwd_io(q, k, v, bias, bias_mode, mask, out, lse, dout, B, H, nq, nk, Dh, dq, dk, dv, dbias, delta,
                           bf16_mma, ldq, ldk, ldv, 0, stream);
}
