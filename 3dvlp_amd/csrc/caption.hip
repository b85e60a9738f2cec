// Caption head kernels (models/caption_module/transformer_captioner.py:286-626 as jointnet.py:104 builds it;
// lib/loss_helper/loss_captioning.py:25-80).
//
// (1) cap_attn: the decoder's self-attention core — 8 heads x 16 channels over <= 64 positions (the reference: 1 object
//     indicator + <= 37 tokens), softmax(q k^T / 4 with masked_fill(mask == 0, -1e9)) -> dropout(0.1) -> v
//     (transformer_captioner.py:32-42).  One wave per (sequence, head): lane = query position, the head's K / V rows sit
//     in LDS and are read as wave-wide broadcasts; backward runs a query pass (dq) and a key pass (dk, dv) in the same
//     wave, so no atomics and no second launch.  The dropout mask is the add & norm hash (common.h), never stored.
//
// (2) vocab_ce: generator + log-softmax + cross entropy without the logits.  The reference materialises
//     log_softmax(proj(x)) as (64, 31, 30 522) fp32 = 242 MB (transformer_captioner.py:106-114) and the loss then reads one
//     entry per row plus the arg-max (loss_captioning.py:37-71).  Here the 1984 x 128 x 30 522 product runs on the matrix
//     cores in (word tile) x (row tile) pieces that never leave the registers:
//       fwd    S^T = W_tile X^T (lane = row: softmax statistics, target logit and arg-max are per-lane scalars),
//              per-(row, vocabulary split) partials -> finalize: lse, target logit, arg-max
//       bwd dX recompute S^T, G = coef (softmax - onehot) in the accumulator layout, dX += G^T W_tile with the accumulator
//              as the next product's A operand (no transposes), atomics over the vocabulary splits
//       bwd dW S = X_tile W^T (lane = word), dW_chunk += G^T X_tile the same way, a workgroup owns its 128 words: plain stores
//     BF = 1: operands rounded to bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate / softmax) — the timing configuration;
//     BF = 0: exact fp32 products (v_mfma_f32_32x32x2_f32) — the 1e-4 parity configuration.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ short bf16_bits(float v) {
  __hip_bfloat16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ------------------------------------------------------------------------------------------------------------------
// (1) attention core, d_k = 16
// ------------------------------------------------------------------------------------------------------------------
constexpr int DK = 16;
constexpr int TMAX = 64;

struct CapAttn {
  const float *qkv;  // (n*T, ld): [q | k | v], each H*16 columns
  int ld;
  const unsigned char *kmask;  // (n, T), 1 = the key may be attended; NULL = all
  int n, T, H, causal;
  float p;
  const unsigned long long *seed;
  int call_id;
  float *out;  // (n*T, H*16)
  float *lse;  // (n*H, T)
  const float *dout;
  float *dqkv;  // (n*T, 3*H*16), contiguous
};

__device__ __forceinline__ float dot16(const float4 (&a)[4], const float4 *b) {
  return dot4(a[0], b[0]) + dot4(a[1], b[1]) + dot4(a[2], b[2]) + dot4(a[3], b[3]);
}
__device__ __forceinline__ void axpy16(float (&acc)[DK], float s, const float4 *v) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 t = v[c];
    acc[4 * c] += s * t.x; acc[4 * c + 1] += s * t.y; acc[4 * c + 2] += s * t.z; acc[4 * c + 3] += s * t.w;
  }
}

__global__ __launch_bounds__(256) void cap_attn_fwd_kernel(CapAttn a) {
  __shared__ float4 Ks[4][TMAX][4], Vs[4][TMAX][4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * 4 + wv;
  const bool act = w < (long long)a.n * a.H && lane < a.T;
  const int seq = (int)(w / a.H), h = (int)(w % a.H), D = a.H * DK;
  float4 q[4];
  bool kv = false;
  if (act) {
    const float *row = a.qkv + ((long long)seq * a.T + lane) * a.ld + h * DK;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q[c] = ld4(row + 4 * c);
      Ks[wv][lane][c] = ld4(row + D + 4 * c);
      Vs[wv][lane][c] = ld4(row + 2 * D + 4 * c);
    }
    kv = a.kmask ? a.kmask[(long long)seq * a.T + lane] != 0 : true;
  }
  const unsigned long long valid = __ballot(kv);
  __syncthreads();
  if (!act) return;
  const float scale = 0.25f;
  float m = -INFINITY;
  for (int j = 0; j < a.T; ++j)
    if (((valid >> j) & 1ull) && (!a.causal || j <= lane)) m = fmaxf(m, dot16(q, Ks[wv][j]) * scale);
  const unsigned thresh = (unsigned)(a.p * 16777216.0f);
  const unsigned mix = a.p > 0.f ? seed_mix_of(a.seed, a.call_id) : 0u;
  const float inv_keep = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.0f;
  const unsigned e0 = ((unsigned)w * (unsigned)a.T + (unsigned)lane) * (unsigned)a.T;
  float l = 0.f, acc[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) acc[c] = 0.f;
  for (int j = 0; j < a.T; ++j) {
    if (!(((valid >> j) & 1ull) && (!a.causal || j <= lane))) continue;
    const float e = __expf(dot16(q, Ks[wv][j]) * scale - m);
    l += e;
    const float pe = (a.p > 0.f && !keep_element(mix, e0 + (unsigned)j, thresh)) ? 0.f : e * inv_keep;
    axpy16(acc, pe, Vs[wv][j]);
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  float *o = a.out + ((long long)seq * a.T + lane) * D + h * DK;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    *reinterpret_cast<float4 *>(o + 4 * c) = make_float4(acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv);
  a.lse[w * a.T + lane] = m + __logf(l);
}

__global__ __launch_bounds__(128) void cap_attn_bwd_kernel(CapAttn a) {
  __shared__ float4 Qs[2][TMAX][4], Ks[2][TMAX][4], Vs[2][TMAX][4], Gs[2][TMAX][4];
  __shared__ float Ls[2][TMAX], Ds[2][TMAX];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * 2 + wv;
  const bool act = w < (long long)a.n * a.H && lane < a.T;
  const int seq = (int)(w / a.H), h = (int)(w % a.H), D = a.H * DK;
  float4 q[4], k[4], v[4], g[4];
  bool kv = false;
  float lse_i = 0.f, delta_i = 0.f;
  if (act) {
    const long long r = (long long)seq * a.T + lane;
    const float *row = a.qkv + r * a.ld + h * DK;
    const float *go = a.dout + r * D + h * DK, *oo = a.out + r * D + h * DK;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q[c] = ld4(row + 4 * c); k[c] = ld4(row + D + 4 * c); v[c] = ld4(row + 2 * D + 4 * c); g[c] = ld4(go + 4 * c);
      delta_i += dot4(g[c], ld4(oo + 4 * c));
      Qs[wv][lane][c] = q[c]; Ks[wv][lane][c] = k[c]; Vs[wv][lane][c] = v[c]; Gs[wv][lane][c] = g[c];
    }
    lse_i = a.lse[w * a.T + lane];
    Ls[wv][lane] = lse_i;
    Ds[wv][lane] = delta_i;
    kv = a.kmask ? a.kmask[r] != 0 : true;
  }
  const unsigned long long valid = __ballot(kv);
  __syncthreads();
  if (!act) return;
  const float scale = 0.25f;
  const unsigned thresh = (unsigned)(a.p * 16777216.0f);
  const unsigned mix = a.p > 0.f ? seed_mix_of(a.seed, a.call_id) : 0u;
  const float inv_keep = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.0f;
  const unsigned wT = (unsigned)w * (unsigned)a.T;
  float dq[DK], dk[DK], dv[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) dq[c] = dk[c] = dv[c] = 0.f;
  // query pass: lane = query i
  for (int j = 0; j < a.T; ++j) {
    if (!(((valid >> j) & 1ull) && (!a.causal || j <= lane))) continue;
    const float pr = __expf(dot16(q, Ks[wv][j]) * scale - lse_i);
    const float mk = (a.p > 0.f && !keep_element(mix, (wT + (unsigned)lane) * (unsigned)a.T + (unsigned)j, thresh)) ? 0.f : inv_keep;
    const float ds = pr * (dot16(g, Vs[wv][j]) * mk - delta_i);
    axpy16(dq, ds * scale, Ks[wv][j]);
  }
  // key pass: lane = key j
  if (kv) {
    for (int i = a.causal ? lane : 0; i < a.T; ++i) {
      const float pr = __expf(dot16(k, Qs[wv][i]) * scale - Ls[wv][i]);
      const float mk = (a.p > 0.f && !keep_element(mix, (wT + (unsigned)i) * (unsigned)a.T + (unsigned)lane, thresh)) ? 0.f : inv_keep;
      axpy16(dv, pr * mk, Gs[wv][i]);
      const float ds = pr * (dot16(v, Gs[wv][i]) * mk - Ds[wv][i]);
      axpy16(dk, ds * scale, Qs[wv][i]);
    }
  }
  float *o = a.dqkv + ((long long)seq * a.T + lane) * 3 * D + h * DK;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    *reinterpret_cast<float4 *>(o + 4 * c) = make_float4(dq[4 * c], dq[4 * c + 1], dq[4 * c + 2], dq[4 * c + 3]);
    *reinterpret_cast<float4 *>(o + D + 4 * c) = make_float4(dk[4 * c], dk[4 * c + 1], dk[4 * c + 2], dk[4 * c + 3]);
    *reinterpret_cast<float4 *>(o + 2 * D + 4 * c) = make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// (2) vocabulary cross entropy, d_model = 128
// ------------------------------------------------------------------------------------------------------------------
constexpr int DM = 128;         // d_model
constexpr int LDB = DM + 8;     // bf16 row-major tile: row stride (elements), 272 B: conflict-free ds_read_b128
constexpr int LDF = DM + 4;     // fp32 row-major tile: row stride (floats)
constexpr int LDT = 32 + 4;     // bf16 transposed tile [d][32 + pad]: 72 B rows, 8-byte aligned quads

struct VocabArgs {
  const float *X;      // (R, 128)
  const float *W;      // (V, 128)
  const float *bias;   // (V) or NULL
  const int *target;   // (R)
  long long R;
  int V, nsplit;
  float4 *part;        // fwd: (2 * nsplit, R): {max, sum exp, target logit or -inf, arg-max as int bits}
  const float *lse;    // bwd
  const float *coef;   // bwd: dLoss/d nll per row
  float *dX;           // bwd dX: (R, 128), zeroed by the host wrapper, atomically accumulated
  float *dW, *dbias;   // bwd dW: (V, 128), (V)
};

// One 32 x 128 operand tile in LDS, row-major, in the MFMA operand type; optionally its [column][row] transpose (bf16).
template <bool BF>
struct Tile {
  // bf16: rows[32][LDB] shorts (+ trans[128][LDT] shorts);  fp32: rows[32][LDF] floats
  static constexpr int ROW_BYTES = BF ? 32 * LDB * 2 : 32 * LDF * 4;
  static constexpr int TR_BYTES = BF ? DM * LDT * 2 : 0;
};

// Stage rows [row0, row0 + 32) of a (nrows, 128) fp32 matrix (rows >= nrows: zeros) with 256 threads.
template <bool BF, bool TRANS>
__device__ __forceinline__ void stage_tile(const float *__restrict__ M, long long row0, long long nrows, char *rows, char *trans) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = threadIdx.x + 256 * u, r = idx >> 5, c4 = idx & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < nrows) v = ld4(M + (row0 + r) * DM + 4 * c4);
    if (BF) {
      short *dst = reinterpret_cast<short *>(rows) + r * LDB + 4 * c4;
      bf16x4 pk;
      pk[0] = bf16_bits(v.x); pk[1] = bf16_bits(v.y); pk[2] = bf16_bits(v.z); pk[3] = bf16_bits(v.w);
      *reinterpret_cast<bf16x4 *>(dst) = pk;
      if (TRANS) {
        short *t = reinterpret_cast<short *>(trans);
        t[(4 * c4 + 0) * LDT + r] = pk[0]; t[(4 * c4 + 1) * LDT + r] = pk[1];
        t[(4 * c4 + 2) * LDT + r] = pk[2]; t[(4 * c4 + 3) * LDT + r] = pk[3];
      }
    } else {
      *reinterpret_cast<float4 *>(reinterpret_cast<float *>(rows) + r * LDF + 4 * c4) = v;
    }
  }
}

// The operand a lane keeps for the whole kernel: its row of a (., 128) fp32 matrix, k = 64 h + ... (any bijection of the
// summation index is fine as long as both operands of a product use the same one).
template <bool BF>
struct RowFrag;
template <>
struct RowFrag<true> {
  bf16x8 v[8];
  __device__ __forceinline__ void load(const float *row, int half, bool ok) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (ok) { a = ld4(row + 64 * half + 8 * s); b = ld4(row + 64 * half + 8 * s + 4); }
      v[s][0] = bf16_bits(a.x); v[s][1] = bf16_bits(a.y); v[s][2] = bf16_bits(a.z); v[s][3] = bf16_bits(a.w);
      v[s][4] = bf16_bits(b.x); v[s][5] = bf16_bits(b.y); v[s][6] = bf16_bits(b.z); v[s][7] = bf16_bits(b.w);
    }
  }
};
template <>
struct RowFrag<false> {
  float v[64];
  __device__ __forceinline__ void load(const float *row, int half, bool ok) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) a = ld4(row + 64 * half + 4 * s);
      v[4 * s] = a.x; v[4 * s + 1] = a.y; v[4 * s + 2] = a.z; v[4 * s + 3] = a.w;
    }
  }
};

// acc (32 x 32) = tile (rows of the LDS tile = the product's rows) x frag^T, or frag x tile^T with the roles swapped:
// TILE_IS_A selects which operand slot the LDS tile takes.
template <bool BF, bool TILE_IS_A>
__device__ __forceinline__ f32x16 tile_times_frag(const char *rows, const RowFrag<BF> &f, int r, int half) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if constexpr (BF) {
    const short *base = reinterpret_cast<const short *>(rows) + r * LDB + 64 * half;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bf16x8 t = *reinterpret_cast<const bf16x8 *>(base + 8 * s);
      acc = TILE_IS_A ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(t, f.v[s], acc, 0, 0, 0)
                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v[s], t, acc, 0, 0, 0);
    }
  } else {
    const float *base = reinterpret_cast<const float *>(rows) + r * LDF + 64 * half;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float4 t = ld4(base + 4 * s);
      const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc = TILE_IS_A ? __builtin_amdgcn_mfma_f32_32x32x2f32(tv[e], f.v[4 * s + e], acc, 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_32x32x2f32(f.v[4 * s + e], tv[e], acc, 0, 0, 0);
    }
  }
  return acc;
}

// out[dt] (32 x 32, rows = G's columns, columns = d in [32 dt, 32 dt + 32)) += G^T x tile, G in the accumulator layout
// (column on the lane, rows in the registers), tile rows = G's rows.  bf16: the accumulator is the A operand with the
// permuted k order 16 s + 8 (j >> 2) + 4 h + (j & 3), the tile comes from its [d][row] transpose; fp32: one step per register.
template <bool BF>
__device__ __forceinline__ void gt_times_tile(const f32x16 &g, const char *rows, const char *trans, f32x16 (&out)[4], int r, int half) {
  if constexpr (BF) {
    bf16x8 a[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[s][j] = bf16_bits(g[8 * s + j]);
    const short *t = reinterpret_cast<const short *>(trans);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const short *col = t + (32 * dt + r) * LDT + 4 * half;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(col + 16 * s), hi = *reinterpret_cast<const bf16x4 *>(col + 16 * s + 8);
        bf16x8 b;
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3]; b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b, out[dt], 0, 0, 0);
      }
    }
  } else {
    const float *base = reinterpret_cast<const float *>(rows);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float *trow = base + acc_row(i, half) * LDF + r;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[i], trow[32 * dt], out[dt], 0, 0, 0);
    }
  }
}

__host__ __device__ __forceinline__ int word_tiles(int V) { return (V + 31) / 32; }

// fwd / bwd-dX: grid (row groups of 128, nsplit); a wave owns 32 rows, the workgroup walks the word tiles of its split.
template <bool BF, bool BWD>
__global__ __launch_bounds__(256) void vocab_rows_kernel(VocabArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *rows = smem, *trans = smem + Tile<BF>::ROW_BYTES;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const long long row = (long long)blockIdx.x * 128 + wv * 32 + r;
  const bool rok = row < a.R;
  RowFrag<BF> xf;
  xf.load(a.X + row * DM, half, rok);
  const int tgt = rok ? a.target[row] : -1;
  const int nt = word_tiles(a.V), per = (nt + a.nsplit - 1) / a.nsplit;
  const int t0 = blockIdx.y * per, t1 = min(nt, t0 + per);
  float m = -INFINITY, l = 0.f, tl = -INFINITY, best = -INFINITY;
  int besti = 0x7fffffff;
  float lse = 0.f, cf = 0.f;
  f32x16 dx[4];
  if (BWD) {
    lse = rok ? a.lse[row] : 1e30f;
    cf = rok ? a.coef[row] : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) dx[dt][i] = 0.f;
  }
  for (int t = t0; t < t1; ++t) {
    __syncthreads();
    stage_tile<BF, BWD>(a.W, (long long)t * 32, a.V, rows, trans);
    __syncthreads();
    f32x16 s = tile_times_frag<BF, true>(rows, xf, r, half);  // s[i] = logit(word 32 t + acc_row(i, half), row)
    const int w0 = t * 32 + 4 * half;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (w0 + 8 * q4 + e < a.V) ? a.bias[w0 + 8 * q4 + e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[4 * q4 + e] += bv[e];
    }
    if (!BWD) {
      float tm = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int word = t * 32 + acc_row(i, half);
        if (word >= a.V) s[i] = -INFINITY;
        if (word == tgt) tl = s[i];
        if (s[i] > best) { best = s[i]; besti = word; }  // ascending words per lane: the first maximum wins
        tm = fmaxf(tm, s[i]);
      }
      if (tm > m) { l *= __expf(m - tm); m = tm; }
      if (m > -INFINITY) {
#pragma unroll
        for (int i = 0; i < 16; ++i) l += __expf(s[i] - m);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int word = t * 32 + acc_row(i, half);
        const float pr = word < a.V ? __expf(s[i] - lse) : 0.f;
        s[i] = cf * (pr - (word == tgt ? 1.f : 0.f));
      }
      gt_times_tile<BF>(s, rows, trans, dx, r, half);
    }
  }
  if (!BWD) {
    if (rok) {
      float4 o;
      o.x = m; o.y = l; o.z = tl; o.w = __int_as_float(besti);
      a.part[((long long)blockIdx.y * 2 + half) * a.R + row] = o;
      // `best` travels with the arg-max: finalize compares values, ties -> smaller word
      reinterpret_cast<float *>(a.part + (long long)2 * a.nsplit * a.R)[((long long)blockIdx.y * 2 + half) * a.R + row] = best;
    }
  } else {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long long rr = (long long)blockIdx.x * 128 + wv * 32 + acc_row(i, half);
        if (rr < a.R) unsafeAtomicAdd(a.dX + rr * DM + 32 * dt + r, dx[dt][i]);
      }
  }
}

// per row: combine the 2 * nsplit partials -> lse, nll = lse - target logit, arg-max
__global__ __launch_bounds__(256) void vocab_finalize_kernel(const float4 *__restrict__ part, long long R, int nsplit, int V,
                                                             float *__restrict__ lse, float *__restrict__ nll,
                                                             int *__restrict__ argmax) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= R) return;
  const float *bestv = reinterpret_cast<const float *>(part + (long long)2 * nsplit * R);
  float m = -INFINITY, tl = -INFINITY, best = -INFINITY;
  int besti = 0x7fffffff;
  for (int s = 0; s < 2 * nsplit; ++s) m = fmaxf(m, part[(long long)s * R + row].x);
  float l = 0.f;
  for (int s = 0; s < 2 * nsplit; ++s) {
    const float4 p = part[(long long)s * R + row];
    if (p.x > -INFINITY) l += p.y * __expf(p.x - m);
    tl = fmaxf(tl, p.z);
    const float bv = bestv[(long long)s * R + row];
    const int bi = __float_as_int(p.w);
    if (bv > best || (bv == best && bi < besti)) { best = bv; besti = bi; }
  }
  const float ls = m + __logf(l);
  lse[row] = ls;
  nll[row] = ls - tl;
  argmax[row] = (besti >= 0 && besti < V) ? besti : 0;  // (all-NaN logits leave no maximum: never hand out an index past V)
}

// bwd-dW: grid = word chunks of 128; a wave owns 32 words and walks ALL row tiles.
template <bool BF>
__global__ __launch_bounds__(256) void vocab_words_kernel(VocabArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *rows = smem, *trans = smem + Tile<BF>::ROW_BYTES;
  float *meta = reinterpret_cast<float *>(smem + Tile<BF>::ROW_BYTES + Tile<BF>::TR_BYTES);  // [3][32]: lse, coef, target
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int word = blockIdx.x * 128 + wv * 32 + r;
  const bool wok = word < a.V;
  RowFrag<BF> wf;
  wf.load(a.W + (long long)word * DM, half, wok);
  const float bv = (wok && a.bias) ? a.bias[word] : 0.f;
  f32x16 dw[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dw[dt][i] = 0.f;
  float db = 0.f;
  const long long ntile = (a.R + 31) / 32;
  for (long long t = 0; t < ntile; ++t) {
    __syncthreads();
    stage_tile<BF, true>(a.X, t * 32, a.R, rows, trans);
    if (threadIdx.x < 32) {
      const long long rr = t * 32 + threadIdx.x;
      const bool ok = rr < a.R;
      meta[threadIdx.x] = ok ? a.lse[rr] : 1e30f;
      meta[32 + threadIdx.x] = ok ? a.coef[rr] : 0.f;
      meta[64 + threadIdx.x] = __int_as_float(ok ? a.target[rr] : -1);
    }
    __syncthreads();
    f32x16 s = tile_times_frag<BF, true>(rows, wf, r, half);  // s[i] = logit(row 32 t + acc_row(i, half), word) - bias
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rl = acc_row(i, half);
      const float pr = wok ? __expf(s[i] + bv - meta[rl]) : 0.f;
      const float gv = meta[32 + rl] * (pr - (__float_as_int(meta[64 + rl]) == word ? 1.f : 0.f));
      s[i] = gv;
      db += gv;
    }
    gt_times_tile<BF>(s, rows, trans, dw, r, half);
  }
  db += __shfl_xor(db, 32);
  if (wok && half == 0 && a.dbias) a.dbias[word] = db;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ww = blockIdx.x * 128 + wv * 32 + acc_row(i, half);
      if (ww < a.V) a.dW[(long long)ww * DM + 32 * dt + r] = dw[dt][i];
    }
}

template <bool BF>
size_t vocab_lds(bool trans, bool meta) {
  return (size_t)Tile<BF>::ROW_BYTES + (trans ? Tile<BF>::TR_BYTES : 0) + (meta ? 3 * 32 * sizeof(float) : 0);
}

}  // namespace

// (1) --------------------------------------------------------------------------------------------------------------
extern "C" int vlp3d_cap_attn_fwd(const float *qkv, int ld, const unsigned char *kmask, int n, int T, int H, int causal, float p,
                                  const unsigned long long *seed, int call_id, float *out, float *lse, void *stream) {
  if (!qkv || !out || !lse || n <= 0 || T <= 0 || T > TMAX || H <= 0 || ld < 3 * H * DK || (ld & 3) || p < 0.f || p >= 1.f ||
      (p > 0.f && !seed) || (long long)n * H * T * T >= (1ll << 32))
    return -22;
  CapAttn a{qkv, ld, kmask, n, T, H, causal, p, seed, call_id, out, lse, nullptr, nullptr};
  const long long waves = (long long)n * H;
  hipLaunchKernelGGL(cap_attn_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlp3d_cap_attn_bwd(const float *qkv, int ld, const unsigned char *kmask, int n, int T, int H, int causal, float p,
                                  const unsigned long long *seed, int call_id, const float *out, const float *lse,
                                  const float *dout, float *dqkv, void *stream) {
  if (!qkv || !out || !lse || !dout || !dqkv || n <= 0 || T <= 0 || T > TMAX || H <= 0 || ld < 3 * H * DK || (ld & 3) ||
      p < 0.f || p >= 1.f || (p > 0.f && !seed) || (long long)n * H * T * T >= (1ll << 32))
    return -22;
  CapAttn a{qkv, ld, kmask, n, T, H, causal, p, seed, call_id, const_cast<float *>(out), const_cast<float *>(lse), dout, dqkv};
  const long long waves = (long long)n * H;
  hipLaunchKernelGGL(cap_attn_bwd_kernel, dim3((unsigned)((waves + 1) / 2)), dim3(128), 0, (hipStream_t)stream, a);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

// (2) --------------------------------------------------------------------------------------------------------------
extern "C" int vlp3d_vocab_ce_splits(long long R, int V) {
  // enough workgroups for 256 CUs x 2, but at least 8 word tiles per workgroup
  const long long groups = (R + 127) / 128;
  long long ns = (512 + groups - 1) / groups;
  const int nt = (V + 31) / 32;
  if (ns > nt / 8) ns = nt / 8;
  if (ns < 1) ns = 1;
  return (int)ns;
}

extern "C" long long vlp3d_vocab_ce_partial_bytes(long long R, int V) {
  return (long long)2 * vlp3d_vocab_ce_splits(R, V) * R * (sizeof(float4) + sizeof(float));
}

extern "C" int vlp3d_vocab_ce_fwd(const float *X, const float *W, const float *bias, const int *target, long long R, int V,
                                  int bf16_mma, void *partials, float *lse, float *nll, int *argmax, void *stream) {
  if (!X || !W || !target || !partials || !lse || !nll || !argmax || R <= 0 || V <= 0) return -22;
  VocabArgs a{};
  a.X = X; a.W = W; a.bias = bias; a.target = target; a.R = R; a.V = V; a.nsplit = vlp3d_vocab_ce_splits(R, V);
  a.part = reinterpret_cast<float4 *>(partials);
  const dim3 grid((unsigned)((R + 127) / 128), (unsigned)a.nsplit);
  hipStream_t s = (hipStream_t)stream;
  if (bf16_mma) hipLaunchKernelGGL((vocab_rows_kernel<true, false>), grid, dim3(256), vocab_lds<true>(false, false), s, a);
  else hipLaunchKernelGGL((vocab_rows_kernel<false, false>), grid, dim3(256), vocab_lds<false>(false, false), s, a);
  VLP3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(vocab_finalize_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, a.part, R, a.nsplit, V, lse, nll, argmax);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlp3d_vocab_ce_bwd(const float *X, const float *W, const float *bias, const int *target, const float *lse,
                                  const float *coef, long long R, int V, int bf16_mma, float *dX, float *dW, float *dbias,
                                  void *stream) {
  if (!X || !W || !target || !lse || !coef || R <= 0 || V <= 0) return -22;
  VocabArgs a{};
  a.X = X; a.W = W; a.bias = bias; a.target = target; a.R = R; a.V = V; a.nsplit = vlp3d_vocab_ce_splits(R, V);
  a.lse = lse; a.coef = coef; a.dX = dX; a.dW = dW; a.dbias = dbias;
  hipStream_t s = (hipStream_t)stream;
  if (dX) {
    hipError_t e = vlp3d_zero_words(dX, (size_t)R * DM, s);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((unsigned)((R + 127) / 128), (unsigned)a.nsplit);
    if (bf16_mma) hipLaunchKernelGGL((vocab_rows_kernel<true, true>), grid, dim3(256), vocab_lds<true>(true, false), s, a);
    else hipLaunchKernelGGL((vocab_rows_kernel<false, true>), grid, dim3(256), vocab_lds<false>(true, false), s, a);
    VLP3D_LAUNCH_CHECK();
  }
  if (dW) {
    const dim3 grid((unsigned)((V + 127) / 128));
    if (bf16_mma) hipLaunchKernelGGL((vocab_words_kernel<true>), grid, dim3(256), vocab_lds<true>(true, true), s, a);
    else hipLaunchKernelGGL((vocab_words_kernel<false>), grid, dim3(256), vocab_lds<false>(true, true), s, a);
    VLP3D_LAUNCH_CHECK();
  }
  return 0;
}
