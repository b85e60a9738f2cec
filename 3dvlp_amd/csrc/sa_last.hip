// Backward of the LAST layer of a grouped MLP (pointnet2_modules.py:251-267: conv -> BatchNorm2d -> ReLU -> max over nsample)
// without its pre-activation — bf16 configuration, round 4.
//
// Through the max-pool only ONE sample of a ball carries gradient per channel, so the masked gradient G (rows x C3) of the last
// layer is sparse: G[u][c] = (sel[ball(u)][c] == position(u)) ? g[ball(u)][c] : 0, with g = dP where the pooled output is
// positive (vlp3d_sa_pool_tstats).  Training-mode BatchNorm backward, summed over the w_u copies of a distinct row u
// (csrc/sa_compact.hip), is   dY3_u = k1 (G_u - w_u (k2 + yhat_u k3)) = k1 G_u - w_u (alpha + beta y3_u)   per channel c, with
// alpha = k1 (k2 + k3 nm), beta = k1 k3 rstd (bn5 = [rstd | nm = -mean rstd | k1 = gamma rstd | k2 = mean g | k3 = mean g yhat]).
// The round-3 kernels (csrc/sa_mlp.hip, pooled-gradient loaders of vlp3d_sa_wgrad / vlp3d_sa_bwd_layer) evaluated that per
// element: every 8-column piece of every ROW fetched the ball's pooled gradient, its arg-max bytes and Y3 — 260 MB of L2
// traffic for SA1's 16 384 balls of 10 MB, and 2-5 x the time per byte of the same kernels' dense layers (SA1 layer 3:
// 153 + 99 us, layer 2: 30 + 49 us).  But y3 = a2 W3^T (a2 = relu(bn2(Y2)), the layer's own input), so
//   dA2 = dY3 W3      = (k1 G) W3 - w (u_alpha + a2 Q),        u_alpha = W3^T alpha,  Q = W3^T diag(beta) W3   (C2 x C2)
//   dW3 = dY3^T a2    = (k1 G)^T a2 - alpha (x) s - diag(beta) W3 M,   s = sum_u w_u a2_u,  M = sum_u w_u a2_u^T a2_u
// and NEITHER needs Y3 (104 MB at SA1, read twice before): both kernels read Y2 once (+ the balls' pooled rows) and run four
// small MFMA products per 32-row tile.  vlp3d_sa_last_dgrad writes the masked gradient of the layer below and its
// BatchNorm-backward sums (the MASK epilogue's contract); vlp3d_sa_last_wgrad writes per-workgroup dW3 slabs with the M / s
// terms already folded in (they are linear in the workgroup's own M, s), so the ordinary slab sum finishes it.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }
__device__ __forceinline__ short bfbits(float v) {
  __hip_bfloat16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}
__device__ __forceinline__ float bf2f(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

struct LastArgs {
  const short *Y2;           // (rows x C2) bf16: pre-activations of the layer below the last
  const float *vec2;         // [4][C2]: scale | shift | rstd | -mean rstd of that layer's BatchNorm (vlp3d_sa_bn_fold)
  const float *bn5;          // [5][C3]: backward constants of the last layer (vlp3d_sa_bn_bwd_consts)
  const short *W3T;          // (C2 x C3) bf16: W3 transposed (vlp3d_sa_prep_weights)
  const short *W3;           // (C3 x C2) bf16: W3 (wgrad only)
  const float *gsel;         // (BM x C3): dP where the pooled output is positive, else 0 (vlp3d_sa_pool_tstats)
  const unsigned char *sel;  // (BM x C3): position of the arg-max sample inside its ball
  const int4 *crow;          // compact row map (csrc/sa_compact.hip) or NULL: dense rows r = bm * S + s, multiplicity 1
  const int *rowptr;
  int nballs, S;
  long long R;               // dense row count B*M*S (a multiple of 32)
  short *G2;                 // dgrad out: (rows x C2) bf16 masked gradient of the layer below
  double *tstats;            // dgrad out: (nslab x 2 x C2) per-workgroup [sum g | sum g yhat] slabs; unused slabs are zeroed
  int nslab;
  float *partials;           // wgrad out: (gridDim.x x C3 x C2) fp32 dW3 slabs
};

__device__ __forceinline__ long long tile_count(const LastArgs &a) {
  return a.crow ? ((long long)a.rowptr[a.nballs] + 31) / 32 : a.R / 32;
}

// (ball << 8 | position, multiplicity) of the 32 rows of tile t -> LDS, by threads 0..31
__device__ __forceinline__ void load_meta(const LastArgs &a, long long t, int *s_meta, float *s_w) {
  if (threadIdx.x < 32) {
    const long long row = t * 32 + threadIdx.x;
    if (a.crow) {
      const int4 cr = a.crow[row];
      s_meta[threadIdx.x] = cr.y;
      s_w[threadIdx.x] = __int_as_float(cr.z);
    } else {
      const int bm = (int)(row / a.S);
      s_meta[threadIdx.x] = (bm << 8) | (int)(row - (long long)bm * a.S);
      s_w[threadIdx.x] = 1.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad: G2 = relu-mask( (k1 G) W3 - w (u_alpha + a2 Q) ), + the BatchNorm-backward sums of the layer below
// ---------------------------------------------------------------------------------------------------------------------
template <int C2, int C3>
__global__ __launch_bounds__(256) void sa_last_dgrad_kernel(LastArgs a) {
  constexpr int LW = C3 + 8, LQ = C2 + 8;     // LDS row strides (shorts): +16 bytes keeps 16-byte fragment reads conflict-free
  constexpr int NCT = C2 / 32;                // 32-column output tiles: wave ct owns tile ct
  constexpr int CY = C2 / 8, CG = C3 / 8;     // 16-byte chunks per row
  constexpr int NY = 32 * CY / 256 > 0 ? 32 * CY / 256 : 1, NG = 32 * CG / 256;
  static_assert(256 % CY == 0 && 256 % CG == 0 && C3 <= 256 && NCT <= 4, "shape");
  extern __shared__ __attribute__((aligned(16))) short lds[];
  short *sW3T = lds;                    // [C2][LW]
  short *sQ = sW3T + C2 * LW;           // [C2][LQ]
  short *sG = sQ + C2 * LQ;             // [32][LW]   k1 G tile
  short *sA = sG + 32 * LW;             // [32][LQ]   a2 tile
  short *sY = sA + 32 * LQ;             // [32][LQ]   raw Y2 tile, then the masked gradient on its way out
  float *s_alpha = reinterpret_cast<float *>(sY + 32 * LQ);  // [C3]
  float *s_beta = s_alpha + C3;                               // [C3]
  float *s_ua = s_beta + C3;                                  // [C2]
  float *s_w = s_ua + C2;                                     // [2][32]
  int *s_meta = reinterpret_cast<int *>(s_w + 64);            // [2][32]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- per-workgroup constants: W3^T, alpha / beta, u_alpha = W3^T alpha, Q = W3^T diag(beta) W3 ----
  for (int c = tid; c < C2 * CG; c += 256) {
    const int row = c / CG, ch = c - row * CG;
    *reinterpret_cast<uint4 *>(sW3T + row * LW + ch * 8) = *reinterpret_cast<const uint4 *>(a.W3T + (size_t)row * C3 + ch * 8);
  }
  if (tid < C3) {
    const float rstd = a.bn5[tid], nm = a.bn5[C3 + tid], k1 = a.bn5[2 * C3 + tid], k2 = a.bn5[3 * C3 + tid], k3 = a.bn5[4 * C3 + tid];
    s_alpha[tid] = k1 * (k2 + k3 * nm);
    s_beta[tid] = k1 * k3 * rstd;
  }
  __syncthreads();
  if (tid < C2) {
    float u = 0.f;
    for (int c = 0; c < C3; ++c) u += s_alpha[c] * bf2f(sW3T[tid * LW + c]);
    s_ua[tid] = u;
  }
  {
    // Q[k][k'] = sum_c (beta_c W3T[k][c]) W3T[k'][c]: the scaled operand goes through the (still unused) tile regions in
    // chunks of 64 channels; tile (i, j) of Q on wave (i * NCT + j) % 4
    constexpr int KCH = 64, LB = KCH + 8;
    short *sWb = sG;  // [C2][LB]: C2 * 72 shorts <= 32 * (LW + 2 LQ) for every supported shape
    static_assert(C2 * LB <= 32 * (LW + 2 * LQ), "scratch");
    constexpr int NQ = (NCT * NCT + 3) / 4;
    f32x16 q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = zero16();
    for (int c0 = 0; c0 < C3; c0 += KCH) {
      __syncthreads();
      for (int e = tid; e < C2 * KCH; e += 256) {
        const int k = e / KCH, c = e - k * KCH;
        sWb[k * LB + c] = bfbits(s_beta[c0 + c] * bf2f(sW3T[k * LW + c0 + c]));
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int t = wave + 4 * i;
        if (t < NCT * NCT) {
          const int ti = t / NCT, tj = t - ti * NCT;
#pragma unroll
          for (int s = 0; s < KCH / 16; ++s) {
            const bf16x8 av = *reinterpret_cast<const bf16x8 *>(sWb + (32 * ti + r) * LB + 16 * s + 8 * half);
            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(sW3T + (32 * tj + r) * LW + c0 + 16 * s + 8 * half);
            q[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, q[i], 0, 0, 0);
          }
        }
      }
    }
    // sQ[k'][k] (B operand of a2 Q: output column k', reduction index k); Q is symmetric up to rounding
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int t = wave + 4 * i;
      if (t < NCT * NCT) {
        const int ti = t / NCT, tj = t - ti * NCT;
#pragma unroll
        for (int e = 0; e < 16; ++e) sQ[(32 * tj + r) * LQ + 32 * ti + acc_row(e, half)] = bfbits(q[i][e]);
      }
    }
  }
  __syncthreads();

  // per-thread constants of the staging columns (a thread always stages the same 8 columns) and of the epilogue column
  const int chy = tid % CY, chg = tid % CG;
  float ysc[8], ysh[8], gk1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ysc[j] = a.vec2[chy * 8 + j];
    ysh[j] = a.vec2[C2 + chy * 8 + j];
    gk1[j] = a.bn5[2 * C3 + chg * 8 + j];
  }
  const int ecol = 32 * (wave < NCT ? wave : 0) + r;
  const float e_sc = a.vec2[ecol], e_sh = a.vec2[C2 + ecol], e_rs = a.vec2[2 * C2 + ecol], e_nm = a.vec2[3 * C2 + ecol];
  const float e_ua = s_ua[ecol];
  double s1 = 0.0, s2 = 0.0;

  // Software pipeline (a tile costs two dependent memory round trips — row words, then the balls' pooled rows — and a
  // workgroup with 100 KB of LDS has the CU to itself): the row words run TWO tiles ahead (registers -> a two-deep LDS
  // ring), the raw Y2 chunks and pooled rows ONE tile ahead (registers), so both latencies pass under the previous tile's
  // products and epilogue.
  const long long ntiles = tile_count(a);
  const long long step = gridDim.x;
  int mreg = 0;
  float wreg = 0.f;
  auto fetch_meta = [&](long long t) {  // threads 0..31: (ball << 8 | position, multiplicity) of row t * 32 + tid
    if (tid < 32 && t < ntiles) {
      const long long row = t * 32 + tid;
      if (a.crow) {
        const int4 cr = a.crow[row];
        mreg = cr.y;
        wreg = __int_as_float(cr.z);
      } else {
        const int bm = (int)(row / a.S);
        mreg = (bm << 8) | (int)(row - (long long)bm * a.S);
        wreg = 1.f;
      }
    }
  };
  uint4 yraw[NY];
  float4 g0r[NG], g1r[NG];
  uint2 slr[NG];
  int posr[NG];
  auto fetch_tile = [&](long long t, int slot) {  // needs s_meta[slot] of tile t
    if (t >= ntiles) return;
    const long long row0 = t * 32;
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int c = tid + 256 * j;
      const int row = (c < 32 * CY ? c : 0) / CY;
      yraw[j] = *reinterpret_cast<const uint4 *>(a.Y2 + (size_t)(row0 + row) * C2 + chy * 8);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = tid + 256 * j, row = c / CG;
      const int meta = s_meta[32 * slot + row], bm = meta >> 8;
      posr[j] = meta & 255;
      g0r[j] = *reinterpret_cast<const float4 *>(a.gsel + (size_t)bm * C3 + chg * 8);
      g1r[j] = *reinterpret_cast<const float4 *>(a.gsel + (size_t)bm * C3 + chg * 8 + 4);
      slr[j] = *reinterpret_cast<const uint2 *>(a.sel + (size_t)bm * C3 + chg * 8);
    }
  };
  long long t = blockIdx.x;
  int slot = 0;
  fetch_meta(t);
  if (tid < 32) { s_meta[tid] = mreg; s_w[tid] = wreg; }
  __syncthreads();
  fetch_tile(t, 0);
  fetch_meta(t + step);
  for (; t < ntiles; t += step, slot ^= 1) {
    const long long row0 = t * 32;
    // registers -> LDS tiles: raw Y2 (the epilogue's mask / yhat), a2 = relu(y scale + shift), k1 G
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int c = tid + 256 * j;
      if (c < 32 * CY) {
        const int row = c / CY;
        *reinterpret_cast<uint4 *>(sY + row * LQ + chy * 8) = yraw[j];
        const short *ys = reinterpret_cast<const short *>(&yraw[j]);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = bfbits(fmaxf(bf2f(ys[k]) * ysc[k] + ysh[k], 0.f));
        *reinterpret_cast<bf16x8 *>(sA + row * LQ + chy * 8) = o;
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = tid + 256 * j, row = c / CG;
      const float gv[8] = {g0r[j].x, g0r[j].y, g0r[j].z, g0r[j].w, g1r[j].x, g1r[j].y, g1r[j].z, g1r[j].w};
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int sk = (int)(((k < 4 ? slr[j].x : slr[j].y) >> (8 * (k & 3))) & 255u);
        o[k] = sk == posr[j] ? bfbits(gk1[k] * gv[k]) : (short)0;
      }
      *reinterpret_cast<bf16x8 *>(sG + row * LW + chg * 8) = o;
    }
    if (tid < 32) { s_meta[32 * (slot ^ 1) + tid] = mreg; s_w[32 * (slot ^ 1) + tid] = wreg; }  // row words of the next tile
    __syncthreads();
    fetch_tile(t + step, slot ^ 1);      // in flight during the products below
    fetch_meta(t + 2 * step);
    if (wave < NCT) {
      f32x16 acc1 = zero16(), acc2 = zero16();
      const short *pg = sG + r * LW + 8 * half, *pw = sW3T + (32 * wave + r) * LW + 8 * half;
#pragma unroll 4
      for (int s = 0; s < C3 / 16; ++s)
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(pg + 16 * s),
                                                       *reinterpret_cast<const bf16x8 *>(pw + 16 * s), acc1, 0, 0, 0);
      const short *pa = sA + r * LQ + 8 * half, *pq = sQ + (32 * wave + r) * LQ + 8 * half;
#pragma unroll 4
      for (int s = 0; s < C2 / 16; ++s)
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(pa + 16 * s),
                                                       *reinterpret_cast<const bf16x8 *>(pq + 16 * s), acc2, 0, 0, 0);
      float ps = 0.f, pq2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = acc_row(i, half);
        short *cell = sY + row * LQ + ecol;
        const float y = bf2f(*cell);
        const float d = acc1[i] - s_w[32 * slot + row] * (e_ua + acc2[i]);
        const float g = (y * e_sc + e_sh > 0.f) ? d : 0.f;
        *cell = bfbits(g);
        ps += g;
        pq2 += g * (y * e_rs + e_nm);
      }
      s1 += (double)ps;
      s2 += (double)pq2;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int c = tid + 256 * j;
      if (c < 32 * CY) {
        const int row = c / CY;
        *reinterpret_cast<uint4 *>(a.G2 + (size_t)(row0 + row) * C2 + chy * 8) = *reinterpret_cast<const uint4 *>(sY + row * LQ + chy * 8);
      }
    }
    __syncthreads();
  }
  // ---- this workgroup's [sum g | sum g yhat] slab; slabs beyond the grid are zeroed ----
  if (wave < NCT) {
    const double t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
    if (half == 0) {
      double *slab = a.tstats + (size_t)blockIdx.x * 2 * C2;
      slab[ecol] = t1;
      slab[C2 + ecol] = t2;
    }
  }
  for (int sb = gridDim.x + blockIdx.x; sb < a.nslab; sb += gridDim.x)
    for (int i = tid; i < 2 * C2; i += 256) a.tstats[(size_t)sb * 2 * C2 + i] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad: slab = (k1 G)^T a2 - alpha (x) s - diag(beta) W3 M over this workgroup's tiles; blockIdx.y = 128-channel block of C3
// ---------------------------------------------------------------------------------------------------------------------
template <int C2, int C3>
__global__ __launch_bounds__(256) void sa_last_wgrad_kernel(LastArgs a) {
  constexpr int CB = 128;                      // channels of C3 per workgroup
  constexpr int LT = 32 + 8;                   // row stride of the transposed tiles (shorts)
  constexpr int LQ = C2 + 8;
  constexpr int NKT = C2 / 32, NCTB = CB / 32;
  constexpr int NP = NCTB * NKT / 4;           // P tiles per wave (2 or 4)
  constexpr int NM = (NKT * NKT + 3) / 4;      // M tiles per wave (1 or 4)
  constexpr int CY = C2 / 8, CG = CB / 8;      // 8-column chunks per row
  constexpr int NY = CY / 8, NG = CG / 8;      // chunks per thread: chunk column = tid / 32 + 8 j, row = tid % 32
  static_assert(C3 % CB == 0 && CY % 8 == 0, "shape");
  constexpr int TILE_SHORTS = (CB + 2 * C2) * LT, TAIL_SHORTS = (CB + 2 * C2) * LQ;
  constexpr int REGION = TILE_SHORTS > TAIL_SHORTS ? TILE_SHORTS : TAIL_SHORTS;
  extern __shared__ __attribute__((aligned(16))) short lds[];
  short *sGt = lds;                      // [CB][LT]  (k1 G)^T           } the tiles of the main loop;
  short *sAt = sGt + CB * LT;            // [C2][LT]  a2^T               } the epilogue's W3 block / M^T images
  short *sAwt = sAt + C2 * LT;           // [C2][LT]  (w a2)^T           } alias them afterwards
  float *s_w = reinterpret_cast<float *>(lds + REGION);     // [2][32]
  int *s_meta = reinterpret_cast<int *>(s_w + 64);           // [2][32]
  float *s_red = reinterpret_cast<float *>(s_meta + 64);     // [256][8 NY] column sums of w a2 per staging thread
  float *s_s = s_red + 256 * 8 * NY;                         // [C2]
  float *s_ab = s_s + C2;                                    // [2][CB]: alpha | beta of this workgroup's channels
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb0 = blockIdx.y * CB;
  // a thread stages row (tid % 32) of the tile, 8-column chunks (tid / 32 + 8 j): its transposed 2-byte LDS writes then run
  // down consecutive shorts of one image row per half-wave (the chunk-fastest mapping put 64 lanes on four banks)
  const int srow = tid & 31, sch = tid >> 5;
  float ysc[NY][8], ysh[NY][8], gk1[NG][8], ssum[NY][8];
#pragma unroll
  for (int j = 0; j < NY; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ysc[j][k] = a.vec2[(sch + 8 * j) * 8 + k];
      ysh[j][k] = a.vec2[C2 + (sch + 8 * j) * 8 + k];
      ssum[j][k] = 0.f;
    }
#pragma unroll
  for (int j = 0; j < NG; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) gk1[j][k] = a.bn5[2 * C3 + cb0 + (sch + 8 * j) * 8 + k];
  f32x16 P[NP], M[NM];
#pragma unroll
  for (int i = 0; i < NP; ++i) P[i] = zero16();
#pragma unroll
  for (int i = 0; i < NM; ++i) M[i] = zero16();

  const long long ntiles = tile_count(a);
  const long long step = gridDim.x;
  int mreg = 0;
  float wreg = 0.f;
  auto fetch_meta = [&](long long t) {
    if (tid < 32 && t < ntiles) {
      const long long row = t * 32 + tid;
      if (a.crow) {
        const int4 cr = a.crow[row];
        mreg = cr.y;
        wreg = __int_as_float(cr.z);
      } else {
        const int bm = (int)(row / a.S);
        mreg = (bm << 8) | (int)(row - (long long)bm * a.S);
        wreg = 1.f;
      }
    }
  };
  uint4 yraw[NY];
  float4 g0r[NG], g1r[NG];
  uint2 slr[NG];
  int posr = 0;
  auto fetch_tile = [&](long long t, int slot) {  // needs s_meta[slot] of tile t
    if (t >= ntiles) return;
    const long long row0 = t * 32;
#pragma unroll
    for (int j = 0; j < NY; ++j)
      yraw[j] = *reinterpret_cast<const uint4 *>(a.Y2 + (size_t)(row0 + srow) * C2 + (sch + 8 * j) * 8);
    const int meta = s_meta[32 * slot + srow], bm = meta >> 8;
    posr = meta & 255;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const size_t at = (size_t)bm * C3 + cb0 + (sch + 8 * j) * 8;
      g0r[j] = *reinterpret_cast<const float4 *>(a.gsel + at);
      g1r[j] = *reinterpret_cast<const float4 *>(a.gsel + at + 4);
      slr[j] = *reinterpret_cast<const uint2 *>(a.sel + at);
    }
  };
  long long t = blockIdx.x;
  int slot = 0;
  fetch_meta(t);
  if (tid < 32) { s_meta[tid] = mreg; s_w[tid] = wreg; }
  __syncthreads();
  fetch_tile(t, 0);
  fetch_meta(t + step);
  for (; t < ntiles; t += step, slot ^= 1) {
    const float w = s_w[32 * slot + srow];
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const short *ys = reinterpret_cast<const short *>(&yraw[j]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const short vb = bfbits(fmaxf(bf2f(ys[k]) * ysc[j][k] + ysh[j][k], 0.f));
        const float wa = w * bf2f(vb);
        sAt[((sch + 8 * j) * 8 + k) * LT + srow] = vb;
        sAwt[((sch + 8 * j) * 8 + k) * LT + srow] = bfbits(wa);
        ssum[j][k] += wa;
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const float gv[8] = {g0r[j].x, g0r[j].y, g0r[j].z, g0r[j].w, g1r[j].x, g1r[j].y, g1r[j].z, g1r[j].w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int sk = (int)(((k < 4 ? slr[j].x : slr[j].y) >> (8 * (k & 3))) & 255u);
        sGt[((sch + 8 * j) * 8 + k) * LT + srow] = sk == posr ? bfbits(gk1[j][k] * gv[k]) : (short)0;
      }
    }
    if (tid < 32) { s_meta[32 * (slot ^ 1) + tid] = mreg; s_w[32 * (slot ^ 1) + tid] = wreg; }
    __syncthreads();
    fetch_tile(t + step, slot ^ 1);     // in flight during the products
    fetch_meta(t + 2 * step);
    // P[c][k] += sum_rows (k1 G)[row][c] a2[row][k];  M[k'][k] += sum_rows (w a2)[row][k'] a2[row][k]
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int tl = wave + 4 * i, ct = tl / NKT, kt = tl - ct * NKT;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        P[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(sGt + (32 * ct + r) * LT + 16 * s2 + 8 * half),
                                                       *reinterpret_cast<const bf16x8 *>(sAt + (32 * kt + r) * LT + 16 * s2 + 8 * half),
                                                       P[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int tl = wave + 4 * i;
      if (tl < NKT * NKT) {
        const int ti = tl / NKT, tj = tl - ti * NKT;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          M[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(sAwt + (32 * ti + r) * LT + 16 * s2 + 8 * half),
                                                         *reinterpret_cast<const bf16x8 *>(sAt + (32 * tj + r) * LT + 16 * s2 + 8 * half),
                                                         M[i], 0, 0, 0);
      }
    }
    __syncthreads();  // the products are done with the tiles before the next trip rewrites them
  }
  if (tid < CB) {
    const int c = cb0 + tid;
    const float rstd = a.bn5[c], nm = a.bn5[C3 + c], k1 = a.bn5[2 * C3 + c], k2 = a.bn5[3 * C3 + c], k3 = a.bn5[4 * C3 + c];
    s_ab[tid] = k1 * (k2 + k3 * nm);
    s_ab[CB + tid] = k1 * k3 * rstd;
  }
  // ---- s = column sums of w a2 over this workgroup's rows (fixed summation order) ----
#pragma unroll
  for (int j = 0; j < NY; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[(tid * NY + j) * 8 + k] = ssum[j][k];
  __syncthreads();
  if (tid < C2) {  // column tid = chunk (tid / 8), element (tid % 8): staged by threads with sch + 8 j == chunk
    const int ch = tid >> 3, k = tid & 7, j = ch >> 3, sc0 = ch & 7;
    float v = 0.f;
    for (int rr = 0; rr < 32; ++rr) v += s_red[((sc0 * 32 + rr) * NY + j) * 8 + k];
    s_s[tid] = v;
  }
  // ---- the M term: D = W3[block] M, with M split into two bf16 parts (its entries are sums over thousands of rows) ----
  short *sW3 = lds;                      // [CB][LQ]
  short *sMh = sW3 + CB * LQ;            // [C2][LQ]: sMh[k][k'] = M[k'][k]  (B operand: output column k)
  short *sMl = sMh + C2 * LQ;
  for (int c = tid; c < CB * CY; c += 256) {
    const int row = c / CY, ch = c - row * CY;
    *reinterpret_cast<uint4 *>(sW3 + row * LQ + ch * 8) = *reinterpret_cast<const uint4 *>(a.W3 + (size_t)(cb0 + row) * C2 + ch * 8);
  }
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    const int tl = wave + 4 * i;
    if (tl < NKT * NKT) {
      const int ti = tl / NKT, tj = tl - ti * NKT;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = M[i][e];
        const short hi = bfbits(v);
        const int at = (32 * tj + r) * LQ + 32 * ti + acc_row(e, half);   // [k = column of M][k' = row of M]
        sMh[at] = hi;
        sMl[at] = bfbits(v - bf2f(hi));
      }
    }
  }
  __syncthreads();
  float *slab = a.partials + (size_t)blockIdx.x * C3 * C2;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int tl = wave + 4 * i, ct = tl / NKT, kt = tl - ct * NKT;
    f32x16 D = zero16();
    const short *pw = sW3 + (32 * ct + r) * LQ + 8 * half;
#pragma unroll 4
    for (int s2 = 0; s2 < C2 / 16; ++s2) {
      const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(pw + 16 * s2);
      D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, *reinterpret_cast<const bf16x8 *>(sMh + (32 * kt + r) * LQ + 16 * s2 + 8 * half), D, 0, 0, 0);
      D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, *reinterpret_cast<const bf16x8 *>(sMl + (32 * kt + r) * LQ + 16 * s2 + 8 * half), D, 0, 0, 0);
    }
    const int k = 32 * kt + r;
    const float sk = s_s[k];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cl = 32 * ct + acc_row(e, half);
      slab[(size_t)(cb0 + cl) * C2 + k] = P[i][e] - s_ab[cl] * sk - s_ab[CB + cl] * D[e];
    }
  }
}

template <int C2, int C3>
size_t dgrad_lds() {
  return (size_t)(C2 * (C3 + 8) + C2 * (C2 + 8) + 32 * (C3 + 8) + 2 * 32 * (C2 + 8)) * 2 + (size_t)(2 * C3 + C2 + 64 + 64) * 4;
}
template <int C2, int C3>
size_t wgrad_lds() {
  const size_t tiles = (size_t)(128 + 2 * C2) * 40 * 2, tail = (size_t)(128 + 2 * C2) * (C2 + 8) * 2;
  return (tiles > tail ? tiles : tail) + (size_t)(64 + 64 + 256 * 8 * (C2 / 64) + C2 + 256) * 4;
}

template <int C2, int C3>
int launch_dgrad(const LastArgs &a, int blocks, hipStream_t s) {
  static std::atomic<unsigned long long> done{0};
  const size_t lds = dgrad_lds<C2, C3>();
  if (lds > 160 * 1024) return VLP3D_EINVAL;
  if (lds > 64 * 1024) {
    const int e = vlp3d_opt_in_lds((const void *)sa_last_dgrad_kernel<C2, C3>, (int)lds, done);
    if (e != VLP3D_OK) return e;
  }
  hipLaunchKernelGGL((sa_last_dgrad_kernel<C2, C3>), dim3(blocks), dim3(256), lds, s, a);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
template <int C2, int C3>
int launch_wgrad(const LastArgs &a, int blocks, hipStream_t s) {
  static std::atomic<unsigned long long> done{0};
  const size_t lds = wgrad_lds<C2, C3>();
  if (lds > 160 * 1024) return VLP3D_EINVAL;
  if (lds > 64 * 1024) {
    const int e = vlp3d_opt_in_lds((const void *)sa_last_wgrad_kernel<C2, C3>, (int)lds, done);
    if (e != VLP3D_OK) return e;
  }
  hipLaunchKernelGGL((sa_last_wgrad_kernel<C2, C3>), dim3(blocks, C3 / 128), dim3(256), lds, s, a);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

bool fill(LastArgs &a, const void *Y2, const float *vec2, const float *bn5, const void *W3T, const void *W3, const float *gsel,
          const unsigned char *sel, long long BM, int S, const void *crow, const int *rowptr, int nballs) {
  if (!Y2 || !vec2 || !bn5 || !gsel || !sel || BM < 1 || S < 1 || S > 255 || ((BM * S) & 31) || BM * S >= (1ll << 31) || BM >= (1 << 23))
    return false;
  if (crow && (!rowptr || nballs != BM)) return false;
  a.Y2 = (const short *)Y2; a.vec2 = vec2; a.bn5 = bn5; a.W3T = (const short *)W3T; a.W3 = (const short *)W3;
  a.gsel = gsel; a.sel = sel; a.crow = (const int4 *)crow; a.rowptr = rowptr; a.nballs = nballs; a.S = S; a.R = BM * S;
  return true;
}

}  // namespace

extern "C" int vlp3d_sa_last_supported(int C2, int C3) {
  return (C2 == 64 && C3 == 128) || (C2 == 128 && C3 == 256) || (C2 == 128 && C3 == 128);
}

// Masked input gradient of the last layer (bf16 storage) + the BatchNorm-backward sums of the layer below: what
// vlp3d_sa_bwd_layer(G = NULL, pool_g, pool_sel) computes, without reading the last layer's pre-activation.  tstats: nslab slabs.
extern "C" int vlp3d_sa_last_dgrad(const void *Y2, const float *vec2, const float *bn5, const void *W3T, const float *gsel,
                                   const unsigned char *sel, long long BM, int S, int C2, int C3, void *G2, double *tstats, int nslab,
                                   const void *crow, const int *rowptr, int nballs, void *stream) {
  LastArgs a = {};
  if (!W3T || !G2 || !tstats || nslab < 1 || !fill(a, Y2, vec2, bn5, W3T, nullptr, gsel, sel, BM, S, crow, rowptr, nballs)) return VLP3D_EINVAL;
  a.G2 = (short *)G2; a.tstats = tstats; a.nslab = nslab;
  const long long tiles = a.R / 32;
  // small weights (SA1): a tile per workgroup round, three workgroups per CU; large ones (100 KB of LDS): one per CU
  static const int cap_small = getenv("VLP3D_SA_LAST_DBLOCKS") ? atoi(getenv("VLP3D_SA_LAST_DBLOCKS")) : 768;
  const int cap = C2 == 64 ? cap_small : 256;
  int blocks = (int)(tiles < cap ? tiles : cap);
  if (blocks > nslab) blocks = nslab;
  if (C2 == 64 && C3 == 128) return launch_dgrad<64, 128>(a, blocks, (hipStream_t)stream);
  if (C2 == 128 && C3 == 256) return launch_dgrad<128, 256>(a, blocks, (hipStream_t)stream);
  if (C2 == 128 && C3 == 128) return launch_dgrad<128, 128>(a, blocks, (hipStream_t)stream);
  return VLP3D_EINVAL;
}

// Weight-gradient slabs of the last layer: partials (blocks x C3 x C2) fp32, to be summed over the blocks (vlp3d_slab_reduce_batch).
extern "C" int vlp3d_sa_last_wgrad(const void *Y2, const float *vec2, const float *bn5, const void *W3, const float *gsel,
                                   const unsigned char *sel, long long BM, int S, int C2, int C3, float *partials, int blocks,
                                   const void *crow, const int *rowptr, int nballs, void *stream) {
  LastArgs a = {};
  if (!W3 || !partials || blocks < 1 || !fill(a, Y2, vec2, bn5, nullptr, W3, gsel, sel, BM, S, crow, rowptr, nballs)) return VLP3D_EINVAL;
  a.partials = partials;
  if (C2 == 64 && C3 == 128) return launch_wgrad<64, 128>(a, blocks, (hipStream_t)stream);
  if (C2 == 128 && C3 == 256) return launch_wgrad<128, 256>(a, blocks, (hipStream_t)stream);
  if (C2 == 128 && C3 == 128) return launch_wgrad<128, 128>(a, blocks, (hipStream_t)stream);
  return VLP3D_EINVAL;
}
