// A chain of nn.Linear stages on 64-row tiles that never leave the CU between the stages (include/vlp3d.h: vlp3d_rows_chain).
//
// Between two attention cores the reference's decoder layer is row-local: fc_o -> dropout -> add -> LayerNorm -> FFN linear1
// -> ReLU -> dropout -> linear2 -> dropout -> add -> LayerNorm -> the next block's projections (attention.py:75,128-130,
// mmattention.py:36-50,84-86).  As separate launches every module costs a kernel boundary, an 8..16 MB store and the same
// bytes loaded again (16 384 rows x 128..256 fp32); the products themselves are 0.5 GF each.  Here a workgroup owns 64 rows:
//   * the current activation tile lives in LDS as bf16 [64][K <= 256] (ping-pong pair), exactly the MFMA operand the unfused
//     kernel (csrc/linear_tile.hip) would have staged from memory;
//   * a stage = for every 128-column block of its weight: stage the [128][128] chunk(s) as bf16, 2 x 2 waves x two 32 x 32
//     accumulators (the linear_tile decomposition), epilogue in registers: + bias, optional store, activation + hash dropout,
//     optional store, bf16 into the other LDS tile;
//   * an add & norm stage routes the fp32 accumulators through LDS (aliasing the weight chunk) and normalises 4 rows per
//     wave at a time, 16 lanes x 8 columns per row: statistics = four 16-lane shuffles, all global accesses 32 bytes per lane;
//   * what backward needs (pre-activations, FFN hidden, xhat, rstd, the stage outputs) is stored on the way — the only global
//     traffic besides the input tile, the residual rows and the weights.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 128;         // reduction chunk = columns per weight block
constexpr int LDW = KC + 8;     // weight chunk row stride (shorts)
constexpr int MAXK = 256;
constexpr int LDX = MAXK + 8;   // activation tile row stride (shorts): 528 B, 16-byte aligned rows, 4 banks shift per row
constexpr int LDF = 132;        // fp32 tile row stride (floats)
constexpr int DLN = 128;        // LayerNorm width
constexpr int SW_BYTES = KC * LDW * 2;           // 34 816 >= 64 * LDF * 4 = 33 792 (the fp32 tile aliases the weight chunk)
static_assert(64 * LDF * 4 <= SW_BYTES, "fp32 tile must fit the weight chunk it aliases");
constexpr int lds_bytes(int tr) { return SW_BYTES + 2 * tr * LDX * 2; }  // 102 400 (TR = 64: one workgroup per CU) / 68 608 (32: two)

struct ChainArgs {
  const void *X;
  int x_bf16;   // X holds bf16 rows (an attention core's output written as bf16): copied to the LDS tile as they are
  long long R;
  const unsigned long long *seed;
  int nstages;
  vlp3d_chain_stage st[VLP3D_CHAIN_MAX_STAGES];
};

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ short bf16_bits(float v) {
  __hip_bfloat16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ bf16x4 pack4(const float4 &v) {
  bf16x4 p;
  p[0] = bf16_bits(v.x); p[1] = bf16_bits(v.y); p[2] = bf16_bits(v.z); p[3] = bf16_bits(v.w);
  return p;
}
__device__ __forceinline__ float act_fwd(float z, int kind) {
  return kind == 0 ? fmaxf(z, 0.f) : 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
}
__device__ __forceinline__ float sum16(float v) {  // over the 16 lanes of a row group
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// TR = rows per workgroup: 64 (waves 2 x 2, two accumulators each) or 32 (waves 1 x 4, one accumulator each; two workgroups
// per CU, whose phases — weight staging, MFMA, row pass — overlap each other instead of waiting at the same barriers)
template <int TR>
__global__ __launch_bounds__(256) void rows_chain_kernel(const ChainArgs a) {
  constexpr int NT = TR / 32;        // accumulators per wave
  constexpr int CW = 32 * NT;        // columns per wave
  constexpr int SX_BYTES = TR * LDX * 2;
  constexpr int RP = TR / 16;        // row passes of the epilogue: 4 rows per wave each
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  short *sW = reinterpret_cast<short *>(smem);
  float *sF = reinterpret_cast<float *>(smem);
  short *cur = reinterpret_cast<short *>(smem + SW_BYTES);
  short *nxt = reinterpret_cast<short *>(smem + SW_BYTES + SX_BYTES);
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = TR == 64 ? (wave & 1) : 0, wc = TR == 64 ? (wave >> 1) : wave;
  const long long row0 = (long long)blockIdx.x * TR;
  const long long R = a.R;

  if (a.x_bf16) {  // (kernel-uniform) the input tile is bf16 already: 16-byte chunks of 8 columns, no conversion
    const int K0 = a.st[0].K, qs = K0 == KC ? 4 : 5, q = 1 << qs;  // chunks per row: 16 or 32
    const int per = TR * q / 256;                                  // 2 .. 8
    const short *Xb = reinterpret_cast<const short *>(a.X);
    bf16x8 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < per) {
        const int e = threadIdx.x + 256 * j, row = e >> qs, c8 = e & (q - 1);
        v[j] = *reinterpret_cast<const bf16x8 *>(Xb + min(row0 + row, R - 1) * K0 + 8 * c8);
      }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < per) {
        const int e = threadIdx.x + 256 * j, row = e >> qs, c8 = e & (q - 1);
        *reinterpret_cast<bf16x8 *>(cur + row * LDX + 8 * c8) = v[j];
      }
  } else {  // the input tile: TR rows x K0 columns, coalesced 16-byte row segments, all loads in flight before the conversions
    const int K0 = a.st[0].K, qs = K0 == KC ? 5 : 6, q = 1 << qs;  // float4 per row: 32 or 64
    const int per = TR * q / 256;                                  // 8 or 16
    const float *Xf = reinterpret_cast<const float *>(a.X);
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < per) {
        const int e = threadIdx.x + 256 * j, row = e >> qs, c4 = e & (q - 1);
        v[j] = ld4(Xf + min(row0 + row, R - 1) * K0 + 4 * c4);
      }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < per) {
        const int e = threadIdx.x + 256 * j, row = e >> qs, c4 = e & (q - 1);
        *reinterpret_cast<bf16x4 *>(cur + row * LDX + 4 * c4) = pack4(v[j]);
      }
  }

  // One flat sequence of [128 x 128] weight chunks over (stage, column block, reduction chunk).  The loads of chunk i+1 are
  // issued right after chunk i went to LDS, so their latency runs under chunk i's MFMA steps and epilogue (as separate
  // phases every chunk exposed one L2 round trip: 8 chunks = a third of the kernel).
  constexpr int NV = KC * (KC / 4) / 256;
  float4 vw[NV];
  int s = 0, cb = 0, k0 = 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
    vw[j] = ld4(a.st[0].W + (long long)col * a.st[0].K + 4 * c4);
  }
  f32x16 acc[NT];
  while (true) {
    const vlp3d_chain_stage &S = a.st[s];
    const int N = S.N, K = S.K;
    const bool feed = s + 1 < a.nstages;  // the result is the next stage's operand
    if (k0 == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    }
    __syncthreads();  // the previous chunk's fragment reads / the previous stage's fp32 tile reads are done
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
      *reinterpret_cast<bf16x4 *>(sW + col * LDW + 4 * c4) = pack4(vw[j]);
    }
    int ns = s, ncb = cb, nk0 = k0 + KC;
    if (nk0 >= K) {
      nk0 = 0;
      ncb = cb + KC;
      if (ncb >= N) {
        ncb = 0;
        ns = s + 1;
      }
    }
    if (ns < a.nstages) {
      const float *wn = a.st[ns].W + (long long)ncb * a.st[ns].K + nk0;
      const int ldn = a.st[ns].K;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
        vw[j] = ld4(wn + (long long)col * ldn + 4 * c4);
      }
    }
    __syncthreads();
    {
      const short *pa = cur + (32 * wr + r) * LDX + k0 + 8 * half;
      const short *pw = sW + (CW * wc + r) * LDW + 8 * half;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        const bf16x8 av = *reinterpret_cast<const bf16x8 *>(pa + 16 * ks);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *reinterpret_cast<const bf16x8 *>(pw + 32 * t * LDW + 16 * ks),
                                                           acc[t], 0, 0, 0);
      }
    }
    if (k0 + KC >= K) {
      // epilogue: accumulators (+ bias) -> fp32 tile over the weight chunk (its last reads end at the barrier) -> row pass,
      // 4 rows per wave at a time, 16 lanes x 8 columns per row: every global access is 32 bytes per lane, 512 per row
      // (stored straight from the accumulators — 4 bytes per lane, 128-byte segments — a 256-column stage ran at 2 TB/s)
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = CW * wc + 32 * t + r;
        const float bv = S.bias ? S.bias[cb + col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sF[(32 * wr + acc_row(i, half)) * LDF + col] = acc[t][i] + bv;
      }
      __syncthreads();
      const int g = lane >> 4, c0 = (lane & 15) * 8;
      const bool ln = S.has_ln != 0, act = S.act_kind >= 0;
      const float p = ln ? S.ln_p : (act ? S.act_p : 0.f);
      const unsigned thresh = (unsigned)(p * 16777216.0f);
      const unsigned mix = p > 0.f ? seed_mix_of(a.seed, ln ? S.ln_call : S.act_call) : 0u;
      const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
      float gam[8], bet[8];
      float4 xr[RP][2];
      if (ln) {
        const float4 g0 = ld4(S.gamma + c0), g1 = ld4(S.gamma + c0 + 4), b0 = ld4(S.beta + c0), b1 = ld4(S.beta + c0 + 4);
        gam[0] = g0.x; gam[1] = g0.y; gam[2] = g0.z; gam[3] = g0.w; gam[4] = g1.x; gam[5] = g1.y; gam[6] = g1.z; gam[7] = g1.w;
        bet[0] = b0.x; bet[1] = b0.y; bet[2] = b0.z; bet[3] = b0.w; bet[4] = b1.x; bet[5] = b1.y; bet[6] = b1.z; bet[7] = b1.w;
#pragma unroll
        for (int it = 0; it < RP; ++it) {  // the residual rows of all passes in flight
          const long long gr = min(row0 + wave * (4 * RP) + it * 4 + g, R - 1);
          xr[it][0] = ld4(S.res + gr * DLN + c0);
          xr[it][1] = ld4(S.res + gr * DLN + c0 + 4);
        }
      }
#pragma unroll
      for (int it = 0; it < RP; ++it) {
        const int row = wave * (4 * RP) + it * 4 + g;
        const long long grow = row0 + row;
        const float4 y0 = *reinterpret_cast<const float4 *>(sF + row * LDF + c0);
        const float4 y1 = *reinterpret_cast<const float4 *>(sF + row * LDF + c0 + 4);
        float y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        const long long o = grow * N + cb + c0;  // element index of y[0]: what the dropout hash counts
        float out[8];
        if (ln) {
          const float x[8] = {xr[it][0].x, xr[it][0].y, xr[it][0].z, xr[it][0].w,
                              xr[it][1].x, xr[it][1].y, xr[it][1].z, xr[it][1].w};
          float v[8], sum = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (p > 0.f) y[j] = keep_element(mix, (unsigned)(o + j), thresh) ? y[j] * inv_keep : 0.f;
            v[j] = x[j] + y[j];
            sum += v[j];
          }
          const float mean = sum16(sum) * (1.0f / DLN);
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = v[j] - mean;
            q += d * d;
          }
          const float rs = rsqrtf(sum16(q) * (1.0f / DLN) + S.eps);
          float h[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            h[j] = (v[j] - mean) * rs;
            out[j] = h[j] * gam[j] + bet[j];
          }
          if (grow < R) {
            float *xh = S.xhat + o, *lo = S.ln_out + o;
            *reinterpret_cast<float4 *>(xh) = make_float4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<float4 *>(xh + 4) = make_float4(h[4], h[5], h[6], h[7]);
            *reinterpret_cast<float4 *>(lo) = make_float4(out[0], out[1], out[2], out[3]);
            *reinterpret_cast<float4 *>(lo + 4) = make_float4(out[4], out[5], out[6], out[7]);
            if ((lane & 15) == 0) S.rstd[grow] = rs;
          }
        } else {
          if (S.v_out && grow < R) {
            if (S.v_out_bf16) {  // the projection an attention core reads next: bf16 rows, what the core rounds to anyway
              bf16x8 vb;
#pragma unroll
              for (int j = 0; j < 8; ++j) vb[j] = bf16_bits(y[j]);
              *reinterpret_cast<bf16x8 *>(reinterpret_cast<short *>(S.v_out) + o) = vb;
            } else {
              *reinterpret_cast<float4 *>(S.v_out + o) = y0;
              *reinterpret_cast<float4 *>(S.v_out + o + 4) = y1;
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = y[j];
            if (act) {
              v = act_fwd(v, S.act_kind);
              if (p > 0.f) v = keep_element(mix, (unsigned)(o + j), thresh) ? v * inv_keep : 0.f;
            }
            out[j] = v;
          }
          if (S.h_out && grow < R) {
            if (S.h_out_bf16) {  // what the next stage multiplies and the layer's weight gradient reads: the bf16 values themselves
              bf16x8 hb;
#pragma unroll
              for (int j = 0; j < 8; ++j) hb[j] = bf16_bits(out[j]);
              *reinterpret_cast<bf16x8 *>(reinterpret_cast<short *>(S.h_out) + o) = hb;
            } else {
              *reinterpret_cast<float4 *>(S.h_out + o) = make_float4(out[0], out[1], out[2], out[3]);
              *reinterpret_cast<float4 *>(S.h_out + o + 4) = make_float4(out[4], out[5], out[6], out[7]);
            }
          }
        }
        if (feed) {
          bf16x8 ob;
#pragma unroll
          for (int j = 0; j < 8; ++j) ob[j] = bf16_bits(out[j]);
          *reinterpret_cast<bf16x8 *>(nxt + row * LDX + cb + c0) = ob;
        }
      }
    }
    if (ns >= a.nstages) break;
    if (ns != s) {  // ordered by the next chunk's first barrier
      short *tmp = cur; cur = nxt; nxt = tmp;
    }
    s = ns; cb = ncb; k0 = nk0;
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// The same tile walk for the gradient: input-gradient products (weights transposed: the same staging and MFMA code) with a
// row pass at every point between them — residual adds, add & norm backward, activation / dropout backward — and the
// gradients the weight-gradient kernels read stored on the way.  The residual gradient of an add & norm waits in registers
// (the row pass always maps a lane to the same rows and columns) until the product chain reaches the tensor it belongs to.
constexpr int BT = 32;                                  // rows per workgroup
constexpr int BRP = BT / 16;                            // row passes: 4 rows per wave each
constexpr int BSX = BT * LDX * 2;
constexpr int BRED = SW_BYTES + 2 * BSX;                // [4][2][128] floats: the waves' dgamma | dbeta sums
constexpr int BWD_LDS = BRED + 4 * 2 * DLN * 4;

struct ChainBwdArgs {
  const float *G;
  long long R;
  const unsigned long long *seed;
  int ngemm;
  vlp3d_chain_bwd_gemm gm[VLP3D_CHAIN_MAX_STAGES];
  vlp3d_chain_bwd_point pt[VLP3D_CHAIN_MAX_STAGES + 1];
};

__device__ __forceinline__ float act_grad(float z, int kind) {
  if (kind == 0) return z > 0.f ? 1.f : 0.f;
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  return cdf + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
}
__device__ __forceinline__ void unpack8(const float4 &a, const float4 &b, float (&v)[8]) {
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float *p, const float (&v)[8]) {
  *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// One point of the chain on the workgroup's rows, columns [cb, cb + 128) of an N-wide gradient.  src: the fp32 tile (row
// stride LDF) or NULL = read G.  dst: the bf16 tile the next product reads, or NULL.
template <bool FROM_G>
__device__ __forceinline__ void bwd_point(const vlp3d_chain_bwd_point &P, const unsigned long long *seed, const float *src,
                                          const float *G, int N, int cb, long long row0, long long R, short *dst,
                                          float (&kept)[BRP][8], float *red, int lane, int wave) {
  const int g = lane >> 4, c0 = (lane & 15) * 8;
  const float p = P.p;
  const unsigned thresh = (unsigned)(p * 16777216.0f);
  const unsigned mix = p > 0.f ? seed_mix_of(seed, P.call) : 0u;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float y[BRP][8], ax[BRP][8], rs[BRP];
  float gam[8], dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dg[j] = db[j] = 0.f;
  if (P.op == 1) unpack8(ld4(P.gamma + c0), ld4(P.gamma + c0 + 4), gam);
#pragma unroll
  for (int it = 0; it < BRP; ++it) {  // every global operand of both passes in flight first
    const int row = wave * (4 * BRP) + it * 4 + g;
    const long long gr = min(row0 + row, R - 1), o = gr * N + cb + c0;
    if (!FROM_G) unpack8(*reinterpret_cast<const float4 *>(src + row * LDF + c0), *reinterpret_cast<const float4 *>(src + row * LDF + c0 + 4), y[it]);
    else unpack8(ld4(G + o), ld4(G + o + 4), y[it]);
    if (P.base) {
      float b[8];
      unpack8(ld4(P.base + o), ld4(P.base + o + 4), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) y[it][j] += b[j];
    }
    if (P.op == 2 && P.aux_bf16) {  // (kernel-uniform) ReLU stage whose forward kept only h = dropout(relu(z)) as bf16 rows:
      // h > 0 <=> z > 0 wherever the mask kept the element, and the mask zeroes the rest either way
      const bf16x8 hb = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const short *>(P.aux) + o);
#pragma unroll
      for (int j = 0; j < 8; ++j) ax[it][j] = __uint_as_float(((unsigned)(unsigned short)hb[j]) << 16);
    } else if (P.op != 0) unpack8(ld4(P.aux + o), ld4(P.aux + o + 4), ax[it]);
    if (P.op == 1) rs[it] = P.rstd[gr];
  }
#pragma unroll
  for (int it = 0; it < BRP; ++it) {
    const int row = wave * (4 * BRP) + it * 4 + g;
    const long long grow = row0 + row, o = grow * N + cb + c0;
    const bool live = grow < R;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = y[it][j] + (P.add_kept ? kept[it][j] : 0.f);
    if (P.op == 1) {
      float gg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gg[j] = v[j] * gam[j];
        if (live) {
          dg[j] += v[j] * ax[it][j];
          db[j] += v[j];
        }
        s1 += gg[j];
        s2 += gg[j] * ax[it][j];
      }
      const float m1 = sum16(s1) * (1.0f / DLN), m2 = sum16(s2) * (1.0f / DLN);
      float dx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dx[j] = rs[it] * (gg[j] - m1 - ax[it][j] * m2);
        v[j] = p > 0.f ? (keep_element(mix, (unsigned)(o + j), thresh) ? dx[j] * inv_keep : 0.f) : dx[j];
        if (P.keep) kept[it][j] = dx[j];
      }
      if (P.dres_out && live) store8(P.dres_out + o, dx);
    } else if (P.op == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[j] * act_grad(ax[it][j], P.act_kind);
        v[j] = p > 0.f ? (keep_element(mix, (unsigned)(o + j), thresh) ? d * inv_keep : 0.f) : d;
      }
    }
    if (P.g_out && live) store8(P.g_out + o, v);
    if (dst) {
      bf16x8 ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) ob[j] = bf16_bits(v[j]);
      *reinterpret_cast<bf16x8 *>(dst + row * LDX + cb + c0) = ob;
    }
  }
  if (P.op == 1 && P.part) {  // (wave-uniform) this workgroup's [sum dout * xhat | sum dout]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dg[j] += __shfl_xor(dg[j], 16);
      dg[j] += __shfl_xor(dg[j], 32);
      db[j] += __shfl_xor(db[j], 16);
      db[j] += __shfl_xor(db[j], 32);
    }
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 2 + 0) * DLN + c0 + j] = dg[j];
        red[(wave * 2 + 1) * DLN + c0 + j] = db[j];
      }
    }
    __syncthreads();
    const int t = threadIdx.x;  // 256 threads = [2][128]
    P.part[(long long)blockIdx.x * 2 * DLN + t] = (red[t] + red[2 * DLN + t]) + (red[4 * DLN + t] + red[6 * DLN + t]);
  }
}

__global__ __launch_bounds__(256) void rows_chain_bwd_kernel(const ChainBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  short *sW = reinterpret_cast<short *>(smem);
  float *sF = reinterpret_cast<float *>(smem);
  short *cur = reinterpret_cast<short *>(smem + SW_BYTES);
  short *nxt = reinterpret_cast<short *>(smem + SW_BYTES + BSX);
  float *red = reinterpret_cast<float *>(smem + BRED);
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long row0 = (long long)blockIdx.x * BT;
  const long long R = a.R;
  constexpr int NV = KC * (KC / 4) / 256;
  float4 vw[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
    vw[j] = ld4(a.gm[0].Wt + (long long)col * a.gm[0].K + 4 * c4);
  }
  float kept[BRP][8];
#pragma unroll
  for (int it = 0; it < BRP; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) kept[it][j] = 0.f;
  for (int cb = 0; cb < a.gm[0].K; cb += KC) bwd_point<true>(a.pt[0], a.seed, nullptr, a.G, a.gm[0].K, cb, row0, R, cur, kept, red, lane, wave);

  int s = 0, cb = 0, k0 = 0;
  f32x16 acc;
  while (true) {
    const vlp3d_chain_bwd_gemm &S = a.gm[s];
    const int N = S.N, K = S.K;
    if (k0 == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    __syncthreads();  // the previous chunk's fragment reads / the previous point's fp32 tile and `red` reads are done
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
      *reinterpret_cast<bf16x4 *>(sW + col * LDW + 4 * c4) = pack4(vw[j]);
    }
    int ns = s, ncb = cb, nk0 = k0 + KC;
    if (nk0 >= K) {
      nk0 = 0;
      ncb = cb + KC;
      if (ncb >= N) {
        ncb = 0;
        ns = s + 1;
      }
    }
    if (ns < a.ngemm) {
      const float *wn = a.gm[ns].Wt + (long long)ncb * a.gm[ns].K + nk0;
      const int ldn = a.gm[ns].K;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
        vw[j] = ld4(wn + (long long)col * ldn + 4 * c4);
      }
    }
    __syncthreads();
    {
      const short *pa = cur + r * LDX + k0 + 8 * half;
      const short *pw = sW + (32 * wave + r) * LDW + 8 * half;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(pa + 16 * ks),
                                                      *reinterpret_cast<const bf16x8 *>(pw + 16 * ks), acc, 0, 0, 0);
    }
    if (k0 + KC >= K) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) sF[acc_row(i, half) * LDF + 32 * wave + r] = acc[i];
      __syncthreads();
      bwd_point<false>(a.pt[s + 1], a.seed, sF, nullptr, N, cb, row0, R, s + 1 < a.ngemm ? nxt : nullptr, kept, red, lane, wave);
    }
    if (ns >= a.ngemm) break;
    if (ns != s) {  // ordered by the next chunk's first barrier
      short *tmp = cur; cur = nxt; nxt = tmp;
    }
    s = ns; cb = ncb; k0 = nk0;
  }
}

}  // namespace

extern "C" int vlp3d_rows_chain_io(const void *X, int x_bf16, long long R, const vlp3d_chain_stage *stages, int nstages,
                                   const unsigned long long *seed, void *stream) {
  if (!X || !stages || R < 1 || nstages < 1 || nstages > VLP3D_CHAIN_MAX_STAGES) return VLP3D_EINVAL;
  ChainArgs a;
  a.X = X;
  a.x_bf16 = x_bf16 != 0;
  a.R = R;
  a.seed = seed;
  a.nstages = nstages;
  for (int s = 0; s < nstages; ++s) {
    const vlp3d_chain_stage &S = stages[s];
    if (!S.W || S.N < KC || S.N % KC || S.K < KC || S.K % KC || S.K > MAXK) return VLP3D_EINVAL;
    if (S.N > (s + 1 < nstages ? MAXK : 3 * KC)) return VLP3D_EINVAL;
    if (s > 0 && S.K != stages[s - 1].N) return VLP3D_EINVAL;
    if (S.has_ln) {
      if (S.N != DLN || S.act_kind >= 0 || !S.res || !S.gamma || !S.beta || !S.ln_out || !S.xhat || !S.rstd) return VLP3D_EINVAL;
      if (S.ln_p < 0.f || S.ln_p >= 1.f || (S.ln_p > 0.f && !seed)) return VLP3D_EINVAL;
    } else if (S.act_kind >= 0) {
      if (S.act_kind > 1 || S.act_p < 0.f || S.act_p >= 1.f || (S.act_p > 0.f && !seed)) return VLP3D_EINVAL;
    }
    if (R * (long long)S.N >= (1ll << 32)) return VLP3D_EINVAL;  // the dropout hash counts elements in 32 bits
    if (S.v_out_bf16 && (S.has_ln || S.act_kind >= 0 || !S.v_out)) return VLP3D_EINVAL;  // plain projections only
    if (S.h_out_bf16 && (S.act_kind < 0 || !S.h_out)) return VLP3D_EINVAL;
    a.st[s] = S;
  }
  static const int tr = getenv("VLP3D_CHAIN_TILE_ROWS") ? atoi(getenv("VLP3D_CHAIN_TILE_ROWS")) : 32;
  static std::atomic<unsigned long long> done64{0}, done32{0};
  if (vlp3d_opt_in_lds(reinterpret_cast<const void *>(rows_chain_kernel<64>), lds_bytes(64), done64) != VLP3D_OK ||
      vlp3d_opt_in_lds(reinterpret_cast<const void *>(rows_chain_kernel<32>), lds_bytes(32), done32) != VLP3D_OK)
    return VLP3D_EINVAL;
  if (tr == 64)
    hipLaunchKernelGGL(rows_chain_kernel<64>, dim3((unsigned)((R + 63) / 64)), dim3(256), lds_bytes(64), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(rows_chain_kernel<32>, dim3((unsigned)((R + 31) / 32)), dim3(256), lds_bytes(32), (hipStream_t)stream, a);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_rows_chain(const float *X, long long R, const vlp3d_chain_stage *stages, int nstages,
                                const unsigned long long *seed, void *stream) {
  return vlp3d_rows_chain_io(X, 0, R, stages, nstages, seed, stream);
}

extern "C" int vlp3d_rows_chain_bwd_blocks(long long R) { return (int)((R + BT - 1) / BT); }

extern "C" int vlp3d_rows_chain_bwd(const float *G, long long R, const vlp3d_chain_bwd_point *points,
                                    const vlp3d_chain_bwd_gemm *gemms, int ngemm, const unsigned long long *seed, void *stream) {
  if (!G || !points || !gemms || R < 1 || ngemm < 1 || ngemm > VLP3D_CHAIN_MAX_STAGES) return VLP3D_EINVAL;
  ChainBwdArgs a;
  a.G = G;
  a.R = R;
  a.seed = seed;
  a.ngemm = ngemm;
  bool kept = false;
  for (int j = 0; j <= ngemm; ++j) {
    const vlp3d_chain_bwd_point &P = points[j];
    const int N = j == 0 ? gemms[0].K : gemms[j - 1].N;
    if (j < ngemm) {
      const vlp3d_chain_bwd_gemm &S = gemms[j];
      if (!S.Wt || S.N < KC || S.N % KC || S.N > MAXK || S.K < KC || S.K % KC || S.K > MAXK) return VLP3D_EINVAL;
      if (j > 0 && S.K != gemms[j - 1].N) return VLP3D_EINVAL;
      a.gm[j] = S;
    } else if (!P.g_out) {
      return VLP3D_EINVAL;
    }
    if (P.op < 0 || P.op > 2 || P.p < 0.f || P.p >= 1.f || (P.p > 0.f && !seed)) return VLP3D_EINVAL;
    if (P.op == 1 && (N != DLN || !P.aux || !P.rstd || !P.gamma)) return VLP3D_EINVAL;
    if (P.op == 2 && (!P.aux || P.act_kind < 0 || P.act_kind > 1)) return VLP3D_EINVAL;
    if (P.aux_bf16 && (P.op != 2 || P.act_kind != 0)) return VLP3D_EINVAL;  // h in place of z: ReLU only
    if (P.op != 1 && (P.keep || P.dres_out || P.part)) return VLP3D_EINVAL;
    if (P.add_kept && (!kept || N != DLN)) return VLP3D_EINVAL;  // nothing kept yet / kept rows are 128 wide
    if (P.op == 1 && P.keep) kept = true;
    if (R * (long long)N >= (1ll << 32)) return VLP3D_EINVAL;
    a.pt[j] = P;
  }
  static std::atomic<unsigned long long> done{0};
  if (vlp3d_opt_in_lds(reinterpret_cast<const void *>(rows_chain_bwd_kernel), BWD_LDS, done) != VLP3D_OK) return VLP3D_EINVAL;
  hipLaunchKernelGGL(rows_chain_bwd_kernel, dim3((unsigned)((R + BT - 1) / BT)), dim3(256), BWD_LDS, (hipStream_t)stream, a);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
