// Grouped per-ball MLP of a set-abstraction layer on the matrix cores — replaces, for
// PointnetSAModuleVotes.forward (lib/pointnet2/pointnet2_modules.py:233-267), the sequence
// group_points x2 -> sub/div -> cat -> 3 x (1x1 conv -> BatchNorm2d -> ReLU) -> max_pool over nsample
// (and its autograd backward), each of which round-trips a (B,C,npoint,nsample) tensor through HBM.
//
// Everything is a product over ROWS r = (b*npoint + m)*nsample + s of row-major matrices:
//     Y_l (R x Cout) = A_{l-1} (R x K) * W_l^T,   A_l = relu(bn_l(Y_l)),   A_0 = gathered [features | xyz | 0]
// One kernel template `row_gemm` does every such product.  A wave owns 32 rows; both MFMA operands are
// 16-byte row slices read straight from memory (lane = row for A, lane = output column for W), so there is
// no LDS staging and no layout shuffle:
//   * the A operand is produced by a LOADER that applies the preceding element-wise stage on the fly:
//       GATHER  rows of the point-major feature tensor through the ball-query indices (+ local xyz),
//       BNRELU  relu(y*scale + shift) of the previous layer's pre-activations (training-mode BN folded into
//               per-channel scale/shift once its batch statistics are known),
//       BNBWD   the BatchNorm backward formula dY = k1*(g - k2 - yhat*k3) from the stored masked gradient g;
//   * the EPILOGUE consumes the accumulators in registers:
//       STORE   write Y_l and accumulate the per-channel sum / sum of squares (BN batch statistics, fp64),
//       MASK    ReLU-mask the propagated gradient with the saved pre-activation, accumulate the two BN-backward
//               reductions, store g_{l-1},
//       SCATTER add the input gradient rows to d(features) (point-major, contiguous atomics) and d(xyz).
// `wgrad` computes dW_l = sum_r dY_l[r]^T A_{l-1}[r] (split over row chunks, fp32 MFMA, coalesced row reads)
// and `pool` takes the max over nsample through the monotone BN+ReLU (max or min of Y by the sign of the scale).
// T = float uses v_mfma_f32_32x32x2_f32 (exact fp32: parity path); T = bf16 uses v_mfma_f32_32x32x16_bf16.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __hip_bfloat16 bf16;

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4(const bf16 *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16 *p) { return __bfloat162float(*p); }
__device__ __forceinline__ void st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16 *p, float v) { *p = __float2bfloat16(v); }
__device__ __forceinline__ short bf16_bits(float v) {
  bf16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}

#ifndef VLP3D_GATHER_PREFETCH
#define VLP3D_GATHER_PREFETCH 1  // hold the NEXT tile's gathered rows in registers across the products (153 -> 114 us at SA1)
#endif
enum Loader { GATHER = 0, BNRELU = 1, BNBWD = 2, PLAIN = 3 };
enum Epilogue { STORE = 0, MASK = 1, SCATTER = 2, BIAS = 3, BIAS_WT = 4 };  // BIAS_WT: BIAS with a K-major weight

// Per-column constants, all fp32 vectors of length >= K (loader) / >= COUT (epilogue).
struct RowGemmArgs {
  // GATHER
  const float *xyz, *new_xyz, *feat_pm;
  const int *idx;
  int N, M, S, C;
  float radius;
  // GATHER, bf16 feature rows (the loader's / the previous level's bf16 copy): (B*N x ldf) bf16, ldf % 8 == 0, columns
  // [C, ldf) zero; NULL = fp32 rows in feat_pm.  The LDS kernels then copy 16-byte chunks without a conversion.
  const void *feat_bf;
  int ldf;
  // BNRELU / BNBWD / PLAIN: source matrices (R x ldin)
  const void *Yin, *Gin;
  int ldin;
  const float *scale, *shift;            // BNRELU: a = relu(y*scale + shift)
  const float *rstd, *nmean_rstd;        // BNBWD: yhat = y*rstd + nmean_rstd
  const float *k1, *k2, *k3;             // BNBWD: dy = k1*(g - k2 - yhat*k3)
  // weight (COUT x K) row-major, K % (8 or 16) == 0
  const void *W;
  int K;
  long long R;
  // STORE
  void *Yout;
  int ldout;
  double *stats;  // [2][COUT]: sum, sum of squares
  // MASK: previous layer's pre-activation (R x ldprev) + its BN scale/shift/rstd/nmean_rstd (length COUT)
  const void *Yprev;
  int ldprev;
  const float *p_scale, *p_shift, *p_rstd, *p_nmean_rstd;
  double *tstats;  // [2][COUT]: sum g, sum g*yhat
  // SCATTER
  float *dfeat_pm, *dxyz, *dnew_xyz;
  // BNBWD of the LAST layer: the masked gradient is synthesised from the pooled tensors (BM x ldin) instead of
  // being read from a dense (R x ldin) matrix: g[r][c] = (sel[bm][c] == s && out[bm][c] > 0) ? dP[bm][c] : 0
  const float *pool_g;   // (BM x ldin): dP where out > 0, else 0 (written by pool_tstats)
  const unsigned char *pool_sel;
  int pool_S, pool_shift;  // pool_shift = log2(pool_S) when it is a power of two, else -1
  int S_shift;             // GATHER: log2(S) when S is a power of two, else -1
  int tile_scene;          // GATHER in row_gemm_lds: 1 when M*S % 32 == 0 (a 32-row tile never straddles two scenes)
  int ldw;                 // BIAS_WT: the weight is given K-major, (K x ldw) row-major (dX = dY W without a transpose)
  // Compact row map (csrc/sa_compact.hip; bf16 LDS kernels only).  crow == NULL: dense rows r = (b*M + m)*S + s.
  // Otherwise the matrices hold the DISTINCT rows of every ball back to back: crow[r] = (global point row, (ball << 8) |
  // position in the ball, float bits of the row's multiplicity w, 0); rowptr[ball] = first compact row of the ball,
  // rowptr[nballs] = number of compact rows (read on the device: the grid is sized for the dense worst case).
  const int4 *crow;
  const int *rowptr;
  int nballs;
};

__device__ __forceinline__ long long compact_tiles(const RowGemmArgs &a) {
  return a.crow ? ((long long)a.rowptr[a.nballs] + 31) / 32 : a.R / 32;
}
__device__ __forceinline__ float row_weight(const RowGemmArgs &a, int row) {
  return a.crow ? __int_as_float(reinterpret_cast<const int *>(a.crow + row)[2]) : 1.f;
}
// (ball, position of the row inside its ball): the pooled-gradient synthesis of the last layer's BatchNorm backward
__device__ __forceinline__ void ball_of_row(const RowGemmArgs &a, int row, int &bm, int &sidx) {
  if (a.crow) {
    const int pk = reinterpret_cast<const int *>(a.crow + row)[1];
    bm = pk >> 8;
    sidx = pk & 255;
  } else {
    bm = a.pool_shift >= 0 ? (row >> a.pool_shift) : (row / a.pool_S);
    sidx = row - bm * a.pool_S;
  }
}

template <typename T, int LOADER>
__device__ __forceinline__ float4 load_a4(const RowGemmArgs &a, long long row, int col0, long long gather_base,
                                          long long xyz_base, long long centre_base) {
  if (LOADER == GATHER) {
    if (col0 < a.C) return ld4(a.feat_pm + gather_base + col0);
    if (col0 == a.C) {
      const float *q = a.xyz + xyz_base;
      const float *c = a.new_xyz + centre_base;
      return make_float4((q[0] - c[0]) / a.radius, (q[1] - c[1]) / a.radius, (q[2] - c[2]) / a.radius, 0.f);
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4 y = ld4(reinterpret_cast<const T *>(a.Yin) + row * a.ldin + col0);
  if (LOADER == PLAIN) return y;
  if (LOADER == BNRELU) {
    const float4 sc = ld4(a.scale + col0), sh = ld4(a.shift + col0);
    return make_float4(fmaxf(0.f, y.x * sc.x + sh.x), fmaxf(0.f, y.y * sc.y + sh.y), fmaxf(0.f, y.z * sc.z + sh.z),
                       fmaxf(0.f, y.w * sc.w + sh.w));
  }
  // BNBWD
  float4 g;
  if (a.pool_g != nullptr) {  // kernel-uniform
    int bm, sidx;
    ball_of_row(a, (int)row, bm, sidx);
    const long long off = (long long)bm * a.ldin + col0;
    const float4 dp = ld4(a.pool_g + off);
    const uchar4 sl = *reinterpret_cast<const uchar4 *>(a.pool_sel + off);
    g = make_float4(sl.x == sidx ? dp.x : 0.f, sl.y == sidx ? dp.y : 0.f, sl.z == sidx ? dp.z : 0.f,
                    sl.w == sidx ? dp.w : 0.f);
  } else {
    g = ld4(reinterpret_cast<const T *>(a.Gin) + row * a.ldin + col0);
  }
  const float4 rs = ld4(a.rstd + col0), nm = ld4(a.nmean_rstd + col0);
  const float4 k1 = ld4(a.k1 + col0), k2 = ld4(a.k2 + col0), k3 = ld4(a.k3 + col0);
  if (a.crow) {  // compact rows: dY summed over the copies of the row, k1 (G - w (k2 + yhat k3))
    const float w = row_weight(a, (int)row);
    return make_float4(k1.x * (g.x - w * (k2.x + (y.x * rs.x + nm.x) * k3.x)), k1.y * (g.y - w * (k2.y + (y.y * rs.y + nm.y) * k3.y)),
                       k1.z * (g.z - w * (k2.z + (y.z * rs.z + nm.z) * k3.z)), k1.w * (g.w - w * (k2.w + (y.w * rs.w + nm.w) * k3.w)));
  }
  return make_float4(k1.x * (g.x - k2.x - (y.x * rs.x + nm.x) * k3.x), k1.y * (g.y - k2.y - (y.y * rs.y + nm.y) * k3.y),
                     k1.z * (g.z - k2.z - (y.z * rs.z + nm.z) * k3.z), k1.w * (g.w - k2.w - (y.w * rs.w + nm.w) * k3.w));
}

// acc[ct] (32 rows x 32 cols) += A(32 x G) * W[32ct..32ct+32][G]^T for one K-group of G columns.
template <typename T, int NCT>
struct Mma;

template <int NCT>
struct Mma<float, NCT> {
  static constexpr int G = 8;  // lane half h covers columns 8g+4h .. +3 (4 k-steps of the 32x32x2 MFMA)
  template <int LOADER, bool WT = false>
  static __device__ __forceinline__ void step(const RowGemmArgs &a, f32x16 (&acc)[NCT], int g, int r, int half,
                                              long long row, long long gb, long long xb, long long cb) {
    const int col0 = 8 * g + 4 * half;
    const float4 av = load_a4<float, LOADER>(a, row, col0, gb, xb, cb);
    const float *W = reinterpret_cast<const float *>(a.W);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      float4 bv;
      if (WT) {  // K-major weight: four dword loads, each coalesced over the 32 output columns of the lanes
        const float *wp = W + (long long)col0 * a.ldw + 32 * ct + r;
        bv = make_float4(wp[0], wp[a.ldw], wp[2 * a.ldw], wp[3 * a.ldw]);
      } else {
        bv = ld4(W + (long long)(32 * ct + r) * a.K + col0);
      }
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[ct], 0, 0, 0);
    }
  }
};

template <int NCT>
struct Mma<bf16, NCT> {
  static constexpr int G = 16;  // lane half h covers columns 16g+8h .. +7 (one 32x32x16 MFMA)
  template <int LOADER, bool WT = false>
  static __device__ __forceinline__ void step(const RowGemmArgs &a, f32x16 (&acc)[NCT], int g, int r, int half,
                                              long long row, long long gb, long long xb, long long cb) {
    static_assert(!WT, "K-major weights are an fp32-only form");
    const int col0 = 16 * g + 8 * half;
    const float4 a0 = load_a4<bf16, LOADER>(a, row, col0, gb, xb, cb);
    const float4 a1 = load_a4<bf16, LOADER>(a, row, col0 + 4, gb, xb, cb);
    bf16x8 av;
    av[0] = bf16_bits(a0.x); av[1] = bf16_bits(a0.y); av[2] = bf16_bits(a0.z); av[3] = bf16_bits(a0.w);
    av[4] = bf16_bits(a1.x); av[5] = bf16_bits(a1.y); av[6] = bf16_bits(a1.z); av[7] = bf16_bits(a1.w);
    const bf16 *W = reinterpret_cast<const bf16 *>(a.W);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(W + (long long)(32 * ct + r) * a.K + col0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[ct], 0, 0, 0);
    }
  }
};

// Per-column reductions of a workgroup -> its own slab [blockIdx.x][2][COUT] (fp64).  Thousands of waves adding
// atomically into the same 2*COUT addresses serialise at the memory side (that alone cost ~0.4 ms per launch at
// R = 1 M); the consumers (bn_fold / bn_bwd_consts) sum the slabs instead.
// slabs the consumers of a statistics buffer sum for R rows = the UNCAPPED grid of the row kernels (vlp3d_sa_stat_slabs)
__host__ __device__ inline unsigned stat_slabs_of(long long R) {
  const long long blocks = (R / 32 + 3) / 4;
  const long long cap = 256 * 4;  // a few persistent workgroups per CU; each wave walks tiles with a grid stride
  return (unsigned)(blocks < cap ? blocks : cap);
}
// A launch with FEWER workgroups than slabs (launch_lds_c caps the grid at one workgroup per CU when only one fits): the
// slabs nobody owns are zeroed by the workgroups that run, slab s by workgroup s mod gridDim.x.
template <int COUT>
__device__ __forceinline__ void zero_unowned_slabs(double *__restrict__ slabs, long long R) {
  const unsigned want = stat_slabs_of(R);
  for (unsigned sl = blockIdx.x + gridDim.x; sl < want; sl += gridDim.x) {
    double *slab = slabs + (size_t)sl * 2 * COUT;
    for (int i = threadIdx.x; i < 2 * COUT; i += 256) slab[i] = 0.0;
  }
}
template <int COUT>
__device__ __forceinline__ void block_stats_to_slab(const double (&s1)[COUT / 32], const double (&s2)[COUT / 32],
                                                    double *__restrict__ slabs, int r, int half, int wave) {
  __shared__ double red[4][2][COUT];
#pragma unroll
  for (int ct = 0; ct < COUT / 32; ++ct) {
    const double t1 = s1[ct] + __shfl_xor(s1[ct], 32);
    const double t2 = s2[ct] + __shfl_xor(s2[ct], 32);
    if (half == 0) {
      red[wave][0][32 * ct + r] = t1;
      red[wave][1][32 * ct + r] = t2;
    }
  }
  __syncthreads();
  double *slab = slabs + (size_t)blockIdx.x * 2 * COUT;
  for (int i = threadIdx.x; i < 2 * COUT; i += 256) {
    const int which = i / COUT, c = i - which * COUT;
    slab[i] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// SCATTER epilogue of the gather layer's backward: acc[ct][i] = element (row trow0 + acc_row(i, half), column 32 ct + r) of
// the input-gradient tile; columns [0,C) -> d(features), [C,C+3) -> d(xyz) / -d(new_xyz), the rest is padding.
// Ball query pads a ball with copies of its FIRST neighbour, so most rows of a ball add into the same point row (SA2 at
// cfg2: 5.7 distinct neighbours of 32), and ALL rows of a ball add into the same centre: those rows are summed inside the
// wave first (16 registers + one cross-half exchange) and leave as ONE atomic per column — 5x fewer memory-side atomics
// at SA2, no 32-way contention on d(new_xyz).  Needs S a power of two >= 16 and whole balls / whole tiles aligned
// (M*S % 32 == 0); other shapes take the row-by-row form.
template <int NCT>
__device__ __forceinline__ void scatter_rows(const RowGemmArgs &a, const f32x16 (&acc)[NCT], int trow0, int r, int half) {
  if (a.crow) {  // compact rows: the copies are already summed, every row carries its global point row and its ball
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int4 cr = a.crow[trow0 + acc_row(i, half)];
      const long long pn = cr.x;
      const int bm = cr.y >> 8;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 32 * ct + r;
        const float v = acc[ct][i];
        if (col < a.C) {
          if (a.dfeat_pm) atomicAdd(a.dfeat_pm + pn * a.C + col, v);
        } else if (col < a.C + 3) {
          const float gv = v / a.radius;
          if (a.dxyz) atomicAdd(a.dxyz + pn * 3 + (col - a.C), gv);
          if (a.dnew_xyz) atomicAdd(a.dnew_xyz + (long long)bm * 3 + (col - a.C), -gv);
        }
      }
    }
    return;
  }
  const int tscene = a.tile_scene ? trow0 / (a.M * a.S) : -1;
  int pr[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pr[i] = a.idx[trow0 + acc_row(i, half)];
  const bool merge = a.S_shift >= 4 && tscene >= 0;
  if (merge) {
    const int groups = a.S >= 32 ? 1 : 2;  // balls per 32-row tile (S = 16: rows 0-15 / 16-31 = registers 0-7 / 8-15)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g >= groups) break;
      const int i0 = groups == 1 ? 0 : 8 * g, i1 = groups == 1 ? 16 : 8 * g + 8;
      const int bm = (trow0 + 16 * g * (groups - 1)) >> a.S_shift;
      const int p0 = a.idx[(long long)bm << a.S_shift];  // the ball's first neighbour = the padding value
      const long long pn0 = (long long)tscene * a.N + p0;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 32 * ct + r;
        float dup = 0.f, all = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < i0 || i >= i1) continue;
          const float v = acc[ct][i];
          all += v;
          if (pr[i] == p0) dup += v;
        }
        dup += __shfl_xor(dup, 32);
        all += __shfl_xor(all, 32);
        if (half == 0) {
          if (col < a.C) {
            if (a.dfeat_pm) atomicAdd(a.dfeat_pm + pn0 * a.C + col, dup);
          } else if (col < a.C + 3) {
            if (a.dxyz) atomicAdd(a.dxyz + pn0 * 3 + (col - a.C), dup / a.radius);
            if (a.dnew_xyz) atomicAdd(a.dnew_xyz + (long long)bm * 3 + (col - a.C), -(all / a.radius));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < i0 || i >= i1 || pr[i] == p0) continue;
        const long long pn = (long long)tscene * a.N + pr[i];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int col = 32 * ct + r;
          const float v = acc[ct][i];
          if (col < a.C) {
            if (a.dfeat_pm) atomicAdd(a.dfeat_pm + pn * a.C + col, v);
          } else if (col < a.C + 3) {
            if (a.dxyz) atomicAdd(a.dxyz + pn * 3 + (col - a.C), v / a.radius);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rr = trow0 + acc_row(i, half);
    const int bm = a.S_shift >= 0 ? (rr >> a.S_shift) : rr / a.S;
    const int b = tscene >= 0 ? tscene : bm / a.M;
    const long long pn = (long long)b * a.N + pr[i];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int col = 32 * ct + r;
      const float v = acc[ct][i];
      if (col < a.C) {
        if (a.dfeat_pm) atomicAdd(a.dfeat_pm + pn * a.C + col, v);
      } else if (col < a.C + 3) {
        const float gv = v / a.radius;
        if (a.dxyz) atomicAdd(a.dxyz + pn * 3 + (col - a.C), gv);
        if (a.dnew_xyz) atomicAdd(a.dnew_xyz + (long long)bm * 3 + (col - a.C), -gv);
      }
    }
  }
}

template <typename T, int COUT, int LOADER, int EPI>
__global__ __launch_bounds__(256) void row_gemm_kernel(RowGemmArgs a) {
  constexpr int NCT = COUT / 32;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long ntiles = compact_tiles(a);  // compact row map: BNBWD loader + SCATTER epilogue only (host-checked)
  if ((EPI == BIAS || EPI == BIAS_WT) && gridDim.y > 1) {  // column-split launch of a small linear layer
    const int c0 = blockIdx.y * COUT;                       // this block owns COUT of the output columns
    a.W = reinterpret_cast<const T *>(a.W) + (EPI == BIAS_WT ? (long long)c0 : (long long)c0 * a.K);
    if (a.scale) a.scale += c0;
    a.Yout = reinterpret_cast<T *>(a.Yout) + c0;
  }

  double s1[NCT], s2[NCT];  // per-lane (column) running reductions of the epilogue
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) s1[ct] = s2[ct] = 0.0;

  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
    const long long row = tile * 32 + r;
    long long gb = 0, xb = 0, cb = 0;
    if (LOADER == GATHER) {
      const long long bm = row / a.S, b = bm / a.M;
      const long long p = a.idx[row];
      gb = (b * a.N + p) * a.C;
      xb = (b * a.N + p) * 3;
      cb = bm * 3;
    }
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = zero16();
    const int ngroups = a.K / Mma<T, NCT>::G;
    for (int g = 0; g < ngroups; ++g)
      Mma<T, NCT>::template step<LOADER, EPI == BIAS_WT>(a, acc, g, r, half, row, gb, xb, cb);

    // ---- epilogue: acc[ct][i] is element (row = tile*32 + acc_row(i,half), col = 32ct + r) ----
    if (EPI == STORE) {
      T *Y = reinterpret_cast<T *>(a.Yout);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[ct][i];
          st1(Y + (tile * 32 + acc_row(i, half)) * a.ldout + 32 * ct + r, v);
          ps += v;
          pq += v * v;
        }
        s1[ct] += (double)ps;
        s2[ct] += (double)pq;
      }
    } else if (EPI == MASK) {
      T *G = reinterpret_cast<T *>(a.Yout);
      const T *Yp = reinterpret_cast<const T *>(a.Yprev);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 32 * ct + r;
        const float sc = a.p_scale[col], sh = a.p_shift[col], rs = a.p_rstd[col], nm = a.p_nmean_rstd[col];
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const long long rr = tile * 32 + acc_row(i, half);
          const float y = ld1(Yp + rr * a.ldprev + col);
          const float g = (y * sc + sh > 0.f) ? acc[ct][i] : 0.f;
          st1(G + rr * a.ldout + col, g);
          ps += g;
          pq += g * (y * rs + nm);
        }
        s1[ct] += (double)ps;
        s2[ct] += (double)pq;
      }
    } else if (EPI == BIAS || EPI == BIAS_WT) {  // plain linear layer: Y = A W^T + bias (bias may be NULL), no statistics
      T *Y = reinterpret_cast<T *>(a.Yout);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const float bv = a.scale ? a.scale[32 * ct + r] : 0.f;  // `scale` doubles as the bias vector here
#pragma unroll
        for (int i = 0; i < 16; ++i)
          st1(Y + (tile * 32 + acc_row(i, half)) * a.ldout + 32 * ct + r, acc[ct][i] + bv);
      }
    } else {  // SCATTER
      scatter_rows<NCT>(a, acc, (int)(tile * 32), r, half);
    }
  }

  if (EPI == STORE || EPI == MASK) block_stats_to_slab<COUT>(s1, s2, (EPI == STORE) ? a.stats : a.tstats, r, half, wave);
}

// Plain linear layer, fp32 in memory, bf16 MFMA operands (rounded in registers), fp32 accumulation: the timing-
// configuration form of row_gemm<float, 32, PLAIN, BIAS / BIAS_WT> (64 exact-fp32 MFMA steps per 32 x 32 x 128 tile -> 8).
// A wave owns one 32 x 32 output tile; grid = (row tiles / 4, N / 32).
template <bool WT>
__global__ __launch_bounds__(256) void linear_bf16_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W,
                                                          int K, int ldw, const float *__restrict__ bias,
                                                          const float *__restrict__ base, long long R,
                                                          float *__restrict__ Y, int ldy) {
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long ntiles = R / 32;
  const int c0 = blockIdx.y * 32;
  const float bv = bias ? bias[c0 + r] : 0.f;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
    const float *xr = X + (tile * 32 + r) * ldx + 8 * half;
    const float *wr = WT ? W + (long long)(8 * half) * ldw + c0 + r : W + (long long)(c0 + r) * K + 8 * half;
    f32x16 acc = zero16();
    const int ng = K / 16;
    // eight K-groups (128 columns) per round: ALL their loads are issued before the first product, so a wave pays one
    // memory latency per 128 columns instead of one per 16 (the rolled loop was 8 dependent round trips at K = 128)
    for (int g0 = 0; g0 < ng; g0 += 8) {
      float4 a0[8], a1[8], b0[8], b1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = min(g0 + u, ng - 1);  // clamped: unconditional loads (the extra groups of a short tail are not used)
        a0[u] = ld4(xr + 16 * g);
        a1[u] = ld4(xr + 16 * g + 4);
        if (WT) {
          const float *wp = wr + (long long)(16 * g) * ldw;
          b0[u] = make_float4(wp[0], wp[ldw], wp[2 * ldw], wp[3 * ldw]);
          b1[u] = make_float4(wp[4 * ldw], wp[5 * ldw], wp[6 * ldw], wp[7 * ldw]);
        } else {
          b0[u] = ld4(wr + 16 * g);
          b1[u] = ld4(wr + 16 * g + 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the loads above the products (the scheduler otherwise re-serialises them)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (g0 + u >= ng) a0[u] = a1[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // branch-free tail: a zero operand adds nothing
        bf16x8 av, bw;
        av[0] = bf16_bits(a0[u].x); av[1] = bf16_bits(a0[u].y); av[2] = bf16_bits(a0[u].z); av[3] = bf16_bits(a0[u].w);
        av[4] = bf16_bits(a1[u].x); av[5] = bf16_bits(a1[u].y); av[6] = bf16_bits(a1[u].z); av[7] = bf16_bits(a1[u].w);
        bw[0] = bf16_bits(b0[u].x); bw[1] = bf16_bits(b0[u].y); bw[2] = bf16_bits(b0[u].z); bw[3] = bf16_bits(b0[u].w);
        bw[4] = bf16_bits(b1[u].x); bw[5] = bf16_bits(b1[u].y); bw[6] = bf16_bits(b1[u].z); bw[7] = bf16_bits(b1[u].w);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc, 0, 0, 0);
      }
    }
    if (base) {  // + a same-shape tensor (the gradient arriving through a residual connection beside the layer)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long long o = (tile * 32 + acc_row(i, half)) * ldy + c0 + r;
        Y[o] = acc[i] + bv + base[o];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) Y[(tile * 32 + acc_row(i, half)) * ldy + c0 + r] = acc[i] + bv;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// row_gemm_lds (bf16): same products, loaders and epilogues as row_gemm, but the operands reach the matrix
// cores through LDS so that every global access is a full, coalesced row segment:
//   * W (COUT x K) is staged once per workgroup;
//   * each wave loads its 32 x K tile cooperatively — consecutive lanes read consecutive 16-byte chunks of a row
//     (the direct form has every lane on a different row: 64 cache lines per load instruction) — applies the
//     loader's element-wise stage, and writes bf16 into its private LDS tile;
//   * MFMA fragments come back with ds_read_b128; rows are padded by 16 bytes, which makes the 16-lane groups of
//     ds_read_b128 hit 16 distinct 4-bank slots (conflict-free for every K used here).
// The tile of a wave is private, so no workgroup barrier is needed inside the tile loop (LDS is in order per wave).
// ------------------------------------------------------------------------------------------------
template <int LOADER>
__device__ __forceinline__ void tile_chunk_load(const RowGemmArgs &a, int row, int col0, float4 &v0, float4 &v1) {
  long long gb = 0, xb = 0, cb = 0;
  if (LOADER == GATHER) {
    if (a.crow) {  // compact rows carry their global point row and their ball
      const int4 cr = a.crow[row];
      gb = (long long)cr.x * a.C;
      xb = (long long)cr.x * 3;
      cb = (long long)(cr.y >> 8) * 3;
    } else {
      const int bm = row / a.S, b = bm / a.M;
      const long long p = a.idx[row];
      gb = ((long long)b * a.N + p) * a.C;
      xb = ((long long)b * a.N + p) * 3;
      cb = (long long)bm * 3;
    }
  }
  v0 = load_a4<bf16, LOADER>(a, row, col0, gb, xb, cb);
  v1 = load_a4<bf16, LOADER>(a, row, col0 + 4, gb, xb, cb);
}

__device__ __forceinline__ uint2 pack4(const float4 &a) {
  union {
    short h[4];
    uint2 u;
  } p;
  p.h[0] = bf16_bits(a.x); p.h[1] = bf16_bits(a.y); p.h[2] = bf16_bits(a.z); p.h[3] = bf16_bits(a.w);
  return p.u;
}

__device__ __forceinline__ uint4 pack8(const float4 &a0, const float4 &a1) {
  union {
    short h[8];
    uint4 u;
  } p;
  p.h[0] = bf16_bits(a0.x); p.h[1] = bf16_bits(a0.y); p.h[2] = bf16_bits(a0.z); p.h[3] = bf16_bits(a0.w);
  p.h[4] = bf16_bits(a1.x); p.h[5] = bf16_bits(a1.y); p.h[6] = bf16_bits(a1.z); p.h[7] = bf16_bits(a1.w);
  return p.u;
}


// ---- 8-column chunk loaders with the per-column constants HOISTED (row_gemm_lds: a lane always works on the same
// 8 columns, chunk = lane % (K/8) with K/8 dividing 64).  Re-loading the 2 (BN+ReLU) or 5 (BN-backward) constant
// vectors for every chunk was 5/7 of the kernel's load instructions and, four chunks in flight, 160 VGPRs.
//   BNRELU: a  = max(0, y*ca + cb)                  ca = scale, cb = shift
//   BNBWD:  dy = k1*(g - k2 - (y*rstd + nmr)*k3)  = ca*g + (cb*y + cc),  ca = k1, cb = -k1*k3*rstd, cc = -k1*(k2 + k3*nmr)
struct Raw8 {
  uint4 y, g;
  float4 dp0, dp1;
  uint2 sel;
};

__device__ __forceinline__ void unpack8(const uint4 &u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// slim operand pair of the cross-tile prefetch (no pooled-gradient fields: those kernels keep the batch form)
template <bool WITH_G>
struct RawPF {
  uint4 y;
};
template <>
struct RawPF<true> {
  uint4 y, g;
};
template <int LOADER>
__device__ __forceinline__ void raw_load_pf(const RowGemmArgs &a, int row, int col0, RawPF<LOADER == BNBWD> &w) {
  w.y = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(a.Yin) + (long long)row * a.ldin + col0);
  if constexpr (LOADER == BNBWD)
    w.g = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(a.Gin) + (long long)row * a.ldin + col0);
}
template <int LOADER>
__device__ __forceinline__ uint4 finish_pf(const RawPF<LOADER == BNBWD> &w, const float (&ca)[8], const float (&cb)[8],
                                           const float (&cc)[8], float mult) {
  float y[8], o[8];
  unpack8(w.y, y);
  if constexpr (LOADER == BNBWD) {
    float g[8];
    unpack8(w.g, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(ca[i], g[i], mult * __builtin_fmaf(cb[i], y[i], cc[i]));
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(0.f, __builtin_fmaf(y[i], ca[i], cb[i]));
  }
  return pack8(make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7]));
}

template <int LOADER>
__device__ __forceinline__ void hoist_consts(const RowGemmArgs &a, int col0, float (&ca)[8], float (&cb)[8], float (&cc)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (LOADER == BNRELU) {
      ca[i] = a.scale[col0 + i];
      cb[i] = a.shift[col0 + i];
      cc[i] = 0.f;
    } else if (LOADER == BNBWD) {
      const float k1 = a.k1[col0 + i], k2 = a.k2[col0 + i], k3 = a.k3[col0 + i];
      ca[i] = k1;
      cb[i] = -(k1 * k3) * a.rstd[col0 + i];
      cc[i] = -k1 * (k2 + k3 * a.nmean_rstd[col0 + i]);
    } else {
      ca[i] = cb[i] = cc[i] = 0.f;
    }
  }
}

// `meta` = ((ball << 8) | position in the ball, multiplicity bits) of the row: the kernel keeps the 32 rows' words of its
// current tile in LDS (requested one tile ahead), so neither the pooled-gradient address nor the multiplicity waits for a
// global read of the row map behind the operand itself (round 3: three dependent round trips per chunk batch -> one).
template <int LOADER>
__device__ __forceinline__ void raw_load8(const RowGemmArgs &a, int row, int col0, int2 meta, Raw8 &w) {
  w.y = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(a.Yin) + (long long)row * a.ldin + col0);
  if (LOADER == BNBWD) {
    if (a.pool_g != nullptr) {  // kernel-uniform
      const long long off = (long long)(meta.x >> 8) * a.ldin + col0;
      w.dp0 = ld4(a.pool_g + off);
      w.dp1 = ld4(a.pool_g + off + 4);
      w.sel = *reinterpret_cast<const uint2 *>(a.pool_sel + off);
    } else {
      w.g = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(a.Gin) + (long long)row * a.ldin + col0);
    }
  }
}

template <int LOADER>
__device__ __forceinline__ uint4 finish8(const RowGemmArgs &a, int2 meta, const Raw8 &w, const float (&ca)[8],
                                         const float (&cb)[8], const float (&cc)[8]) {
  float y[8], o[8];
  unpack8(w.y, y);
  if (LOADER == BNRELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(0.f, __builtin_fmaf(y[i], ca[i], cb[i]));
  } else {
    float g[8];
    const float mult = __int_as_float(meta.y);
    if (a.pool_g != nullptr) {
      const unsigned sidx = (unsigned)(meta.x & 255);
      const float dp[8] = {w.dp0.x, w.dp0.y, w.dp0.z, w.dp0.w, w.dp1.x, w.dp1.y, w.dp1.z, w.dp1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned sb = ((i < 4 ? w.sel.x : w.sel.y) >> (8 * (i & 3))) & 0xffu;
        g[i] = sb == sidx ? dp[i] : 0.f;
      }
    } else {
      unpack8(w.g, g);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(ca[i], g[i], mult * __builtin_fmaf(cb[i], y[i], cc[i]));
  }
  return pack8(make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7]));
}

// second launch bound: at least two waves per SIMD for outputs up to 160 columns (the register allocator otherwise
// settles just above 256 registers for <128, BNBWD, MASK>: one wave per SIMD on a latency-bound kernel)
// WIDE (GATHER only): the instantiation for 160 < K <= 288 — its own kernel so that the 144 operand registers of the wide
// loader and the prefetch registers of the narrow one never meet in one allocation (together: 315 spilled VGPRs)
template <int COUT, int LOADER, int EPI, bool WIDE = false>
__global__ __launch_bounds__(256, (COUT <= 256 ? 2 : 1)) void row_gemm_lds_kernel(RowGemmArgs a) {
  typedef bf16 T;
  constexpr int NCT = COUT / 32;
  constexpr int CC = COUT / 8;  // 16-byte chunks per output row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = a.K, ldw = K + 8, kc = K / 8;  // kc = 16-byte chunks per input row
  const int lde = COUT + 8;                    // padded row of the output staging tile
  const int wtile = 32 * (ldw > lde ? ldw : lde);  // per-wave LDS tile, shared by the A operand and the epilogue
  bf16 *sW = reinterpret_cast<bf16 *>(smem);
  bf16 *sA = sW + (size_t)COUT * ldw + (size_t)wave * wtile;
  // per wave: the row-map words ((ball << 8) | position, multiplicity) of the 32 rows of its CURRENT tile (see raw_load8)
  int2 *sMeta = reinterpret_cast<int2 *>(sW + (size_t)COUT * ldw + (size_t)4 * wtile) + wave * 32;
  const long long ntiles = compact_tiles(a);
  const bool compact = a.crow != nullptr;
  if ((long long)blockIdx.x * 4 >= ntiles) {
    // no tile for any wave of this workgroup (a compact row map shrinks the work under a grid sized for the padded rows):
    // leave before staging the weight; the statistic slab of this workgroup is still part of the sums
    if (EPI == STORE || EPI == MASK) {
      double *slab = ((EPI == STORE) ? a.stats : a.tstats) + (size_t)blockIdx.x * 2 * COUT;
      for (int i = threadIdx.x; i < 2 * COUT; i += 256) slab[i] = 0.0;
      zero_unowned_slabs<COUT>((EPI == STORE) ? a.stats : a.tstats, a.R);
    }
    return;
  }

  {  // stage the weight once per workgroup
    const bf16 *W = reinterpret_cast<const bf16 *>(a.W);
    for (int c = threadIdx.x; c < COUT * kc; c += 256) {
      const int row = c / kc, ch = c - row * kc;
      *reinterpret_cast<uint4 *>(sW + row * ldw + ch * 8) = *reinterpret_cast<const uint4 *>(W + (long long)row * K + ch * 8);
    }
  }
  __syncthreads();

  double s1[NCT], s2[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) s1[ct] = s2[ct] = 0.0;

  const int nch = 32 * kc;  // chunks of this wave's tile; 64 lanes take them 64 at a time, 4 batches in flight
  const unsigned kc_inv = ((1u << 20) + kc - 1) / kc;  // c / kc == (c * kc_inv) >> 20 for c < 32 * kc, kc <= 64
  // kc divides 64 (host-checked): fixed columns per lane.  Wide outputs keep the accumulators' registers instead.
  constexpr bool HOIST = (LOADER == BNRELU || LOADER == BNBWD) && !(LOADER == BNRELU && COUT >= 256);
  float ca[8], cb[8], cc[8];
  hoist_consts<HOIST ? LOADER : GATHER>(a, (lane % kc) * 8, ca, cb, cc);
  // GATHER fast path (tile_scene): the chunk -> (row, column) map of a lane does not depend on the tile, the scene index
  // is a per-tile scalar and the ball-query indices of the NEXT tile are fetched while this one is computed.  The
  // generic path below spends three integer divisions per chunk (27 per lane and tile at SA1 — more VALU work than
  // the tile's matrix products) and waits for idx before it can ask for a feature row.
  constexpr int MAXCH = WIDE ? 1 : 10;  // chunks per lane: 32 * (K/8) / 64, K <= 160
  constexpr bool GATHER_PREFETCH = VLP3D_GATHER_PREFETCH != 0;
  const bool fastg = !WIDE && LOADER == GATHER && (a.tile_scene || compact) && nch <= 64 * MAXCH;
  constexpr int WIDECH = WIDE ? 18 : 1;  // K <= 288
  const bool wideg = WIDE;               // launch_lds_c checked the shape
  int crow[MAXCH], ccol[MAXCH], pidx[MAXCH];
  const long long tile_step = (long long)gridDim.x * 4;
  // The gathered rows of tile t+1 are requested BEFORE tile t goes to the matrix cores and stay in registers (gv0 / gv1)
  // through its products and epilogue; the ball-query indices run one tile further ahead still.  (One tile at a time —
  // request, wait, LDS, products, epilogue — left a wave idle for two memory latencies per tile at two waves per SIMD:
  // 20 us per tile.)
  float4 gv0[MAXCH], gv1[MAXCH];
  // the [dx, dy, dz, 0] columns are written by a separate pass (lane = row, lanes 0..31): inside the chunk loop the
  // three divisions by the radius were a ~40-instruction sequence executed for EVERY chunk batch (some lane of the wave
  // always holds a row's xyz chunk) — a quarter of the kernel's vector instructions
  int prow = 0;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  const float inv_radius = 1.f / a.radius;
  auto src_index = [&](long long row) -> int { return compact ? a.crow[row].x : a.idx[row]; };
  auto gather_issue = [&](long long tile) {  // pidx / prow hold the indices of `tile`
    const int row0 = (int)(tile * 32);
    const int scene = compact ? 0 : row0 / (a.M * a.S);  // wave-uniform (R < 2^31); compact rows carry global point rows
    const float *fbase = a.feat_pm + (long long)scene * a.N * a.C;
    const float *xbase = a.xyz + (long long)scene * a.N * 3;
    if (a.feat_bf != nullptr) {  // kernel-uniform: one 16-byte load per chunk, already bf16
      const bf16 *bbase = reinterpret_cast<const bf16 *>(a.feat_bf) + (long long)scene * a.N * a.ldf;
#pragma unroll
      for (int u = 0; u < MAXCH; ++u) {
        gv0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (64 * u < nch && ccol[u] < a.C) gv0[u] = *reinterpret_cast<const float4 *>(bbase + (long long)pidx[u] * a.ldf + ccol[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < MAXCH; ++u) {
        const float *fr = fbase + (long long)pidx[u] * a.C;
        const int col = ccol[u];
        gv0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        gv1[u] = gv0[u];
        if (64 * u < nch) {  // uniform
          if (col < a.C) gv0[u] = ld4(fr + col);
          if (col + 4 < a.C) gv1[u] = ld4(fr + col + 4);
        }
      }
    }
    const float *q = xbase + (long long)prow * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  };
  auto gather_commit = [&](long long tile) {  // registers -> this wave's LDS tile (bf16)
    const int row0 = (int)(tile * 32);
#pragma unroll
    for (int u = 0; u < MAXCH; ++u)
      if (64 * u + lane < nch)
        *reinterpret_cast<uint4 *>(sA + crow[u] * ldw + ccol[u]) =
            a.feat_bf != nullptr ? *reinterpret_cast<const uint4 *>(&gv0[u]) : pack8(gv0[u], gv1[u]);
    if (lane < 32) {  // same wave, later instruction: lands after the chunk writes above
      const int rr = row0 + lane;
      const int bm = compact ? (sMeta[lane].x >> 8) : (a.S_shift >= 0 ? (rr >> a.S_shift) : rr / a.S);
      const float *cc3 = a.new_xyz + (long long)bm * 3;  // a handful of L1-resident centres per tile
      *reinterpret_cast<uint2 *>(sA + lane * ldw + a.C) =
          pack4(make_float4((qx - cc3[0]) * inv_radius, (qy - cc3[1]) * inv_radius, (qz - cc3[2]) * inv_radius, 0.f));
    }
  };
  // Processing order: plain grid stride over the tiles in row (= FPS) order.  Tried and rejected: walking a scene's
  // balls in Morton order of their centres, alone (133 us vs 114 us for SA1 layer 1) or with each XCD working through one
  // contiguous eighth of that order (125 us) — the gathered rows come from L2 / the Infinity Cache either way, and
  // neighbouring waves asking for the SAME lines at the same time is slower than a random spread.
  const long long vt0 = (long long)blockIdx.x * 4 + wave, vstep = tile_step, vend = ntiles;
  auto map_tile = [&](long long v) -> long long { return v; };
  if (fastg) {
#pragma unroll
    for (int u = 0; u < MAXCH; ++u) {
      const int c = min(64 * u + lane, nch - 1);
      crow[u] = c / kc;
      ccol[u] = (c - crow[u] * kc) * 8;
    }
    if (vt0 < vend) {
      const long long t0 = map_tile(vt0);
#pragma unroll
      for (int u = 0; u < MAXCH; ++u) pidx[u] = src_index(t0 * 32 + crow[u]);
      prow = src_index(t0 * 32 + (lane & 31));
      if (GATHER_PREFETCH) {
        gather_issue(t0);
        const long long t1 = map_tile(vt0 + vstep < vend ? vt0 + vstep : vt0);
#pragma unroll
        for (int u = 0; u < MAXCH; ++u) pidx[u] = src_index(t1 * 32 + crow[u]);
        prow = src_index(t1 * 32 + (lane & 31));
      }
    }
  }
  // cross-tile prefetch buffer of the BN loaders (see the tile loop); pooled-gradient operands keep the batch form
  // (BN-backward operands are twice as wide: four chunks, i.e. K <= 64, or the kernel spills)
  constexpr int HP = (HOIST && LOADER == BNBWD) ? 4 : 8;
  constexpr bool PF = HOIST && !(LOADER == BNBWD && COUT >= 128);  // those kernels sit at the 256-register line already
  const bool hoist_pf = PF && nch <= 64 * HP && a.pool_g == nullptr;
  RawPF<HOIST && LOADER == BNBWD> hp[PF ? HP : 1];
  if (PF && hoist_pf && vt0 < vend) {
    const int kshift = __builtin_ctz(kc);
#pragma unroll
    for (int u = 0; u < (PF ? HP : 1); ++u) {
      const int c = min(64 * u + lane, nch - 1);
      raw_load_pf<HOIST ? LOADER : BNRELU>(a, (int)(vt0 * 32) + (c >> kshift), (c & (kc - 1)) * 8, hp[u]);
    }
  }
  // row-map words of the NEXT tile, one row per lane (lanes 0..31), requested a tile ahead; dense rows synthesise them
  auto meta_of = [&](long long tile) -> int2 {
    const int row = (int)(tile * 32) + (lane & 31);
    if (compact) {
      const int4 c = a.crow[row];
      return make_int2(c.y, c.z);
    }
    int bm = 0, sidx = 0;
    if (LOADER == BNBWD && a.pool_g != nullptr) {
      bm = a.pool_shift >= 0 ? (row >> a.pool_shift) : (row / a.pool_S);
      sidx = row - bm * a.pool_S;
    }
    return make_int2((bm << 8) | sidx, __float_as_int(1.f));
  };
  int2 pmeta = make_int2(0, __float_as_int(1.f));
  if (vt0 < vend) pmeta = meta_of(map_tile(vt0));
  for (long long vt = vt0; vt < vend; vt += vstep) {
    const long long tile = map_tile(vt);
    const int row0 = (int)(tile * 32);
    if (lane < 32) sMeta[lane] = pmeta;  // this wave's earlier reads of the previous tile's words are done (in-order LDS)
    if (vt + vstep < vend) pmeta = meta_of(map_tile(vt + vstep));
    if (fastg) {
      if (!GATHER_PREFETCH) gather_issue(tile);
      gather_commit(tile);
      const long long vnext = vt + vstep;
      if (GATHER_PREFETCH) {
        if (vnext < vend) {
          gather_issue(map_tile(vnext));  // in flight during the MFMAs / epilogue of this tile
          const long long t2 = map_tile(vnext + vstep < vend ? vnext + vstep : vnext);
#pragma unroll
          for (int u = 0; u < MAXCH; ++u) pidx[u] = src_index(t2 * 32 + crow[u]);
          prow = src_index(t2 * 32 + (lane & 31));
        }
      } else {
        const long long t1 = map_tile(vnext < vend ? vnext : vt);
#pragma unroll
        for (int u = 0; u < MAXCH; ++u) pidx[u] = src_index(t1 * 32 + crow[u]);  // in flight during the MFMAs / epilogue
        prow = src_index(t1 * 32 + (lane & 31));
      }
    }
    // BN loaders with K <= 128 (at most 8 chunks per lane): the whole NEXT tile is requested before this tile goes to the
    // matrix cores and waits in registers (hp) through products and epilogue, like the gathered operand above.
    if (PF && hoist_pf) {
      const int kshift = __builtin_ctz(kc);
#pragma unroll
      for (int u = 0; u < (PF ? HP : 1); ++u) {
        const int c = 64 * u + lane;
        if (c < nch) {
          const int row = c >> kshift, ch = c & (kc - 1);
          *reinterpret_cast<uint4 *>(sA + row * ldw + ch * 8) =
              finish_pf<HOIST ? LOADER : BNRELU>(hp[u], ca, cb, cc, LOADER == BNBWD ? __int_as_float(sMeta[row].y) : 1.f);
        }
      }
      const long long vnext = vt + vstep;
      if (vnext < vend) {
        const int nrow0 = (int)(vnext * 32);
#pragma unroll
        for (int u = 0; u < (PF ? HP : 1); ++u) {
          const int c = min(64 * u + lane, nch - 1);
          raw_load_pf<HOIST ? LOADER : BNRELU>(a, nrow0 + (c >> kshift), (c & (kc - 1)) * 8, hp[u]);
        }
      }
    }
    // WIDE gather (160 < K <= 288: the 256-channel levels SA3 / SA4 / vote aggregation, 17 chunks per lane): these layers have
    // one or two tiles per wave, so there is no steady state to prefetch into — what counts is the number of dependent round
    // trips inside ONE tile.  The generic loop below takes the chunks four at a time, each batch waiting for its row-map word
    // and then for its feature row: ten round trips, ~25 us per tile in the step.  Here all row-map words are requested first,
    // then all feature rows (144 registers — the accumulators are not live yet): two round trips.
    if constexpr (LOADER == GATHER && WIDE) {
      const int scene = compact ? 0 : row0 / (a.M * a.S);
      const float *fbase = a.feat_pm + (long long)scene * a.N * a.C;
      int wp[WIDECH];
#pragma unroll
      for (int u = 0; u < WIDECH; ++u) {
        const int c = min(64 * u + lane, nch - 1);
        wp[u] = src_index((long long)row0 + (int)(((unsigned)c * kc_inv) >> 20));
      }
      const int wprow = src_index((long long)row0 + (lane & 31));
      float4 w0[WIDECH], w1[WIDECH];
      const bool wbf = a.feat_bf != nullptr;  // kernel-uniform: bf16 feature rows — one 16-byte load per chunk, stored as it is
      if (wbf) {
        const bf16 *bbase = reinterpret_cast<const bf16 *>(a.feat_bf) + (long long)scene * a.N * a.ldf;
#pragma unroll
        for (int u = 0; u < WIDECH; ++u) {
          const int c = min(64 * u + lane, nch - 1);
          const int rw = (int)(((unsigned)c * kc_inv) >> 20), col = (c - rw * kc) * 8;
          w0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (64 * u < nch && col < a.C) w0[u] = *reinterpret_cast<const float4 *>(bbase + (long long)wp[u] * a.ldf + col);
        }
      } else {
#pragma unroll
        for (int u = 0; u < WIDECH; ++u) {
          const int c = min(64 * u + lane, nch - 1);
          const int rw = (int)(((unsigned)c * kc_inv) >> 20), col = (c - rw * kc) * 8;
          const float *fr = fbase + (long long)wp[u] * a.C;
          w0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          w1[u] = w0[u];
          if (64 * u < nch) {  // uniform
            if (col < a.C) w0[u] = ld4(fr + col);
            if (col + 4 < a.C) w1[u] = ld4(fr + col + 4);
          }
        }
      }
      const float *wq = a.xyz + ((long long)scene * a.N + wprow) * 3;
      const float wx = wq[0], wy = wq[1], wz = wq[2];
#pragma unroll
      for (int u = 0; u < WIDECH; ++u) {
        const int c = 64 * u + lane;
        if (c < nch) {
          const int rw = (int)(((unsigned)c * kc_inv) >> 20), col = (c - rw * kc) * 8;
          *reinterpret_cast<uint4 *>(sA + rw * ldw + col) = wbf ? *reinterpret_cast<const uint4 *>(&w0[u]) : pack8(w0[u], w1[u]);
        }
      }
      if (lane < 32) {  // [dx, dy, dz, 0] columns: same wave, later instruction — lands after the chunk writes above
        const int rr = row0 + lane;
        const int bm = compact ? (sMeta[lane].x >> 8) : (a.S_shift >= 0 ? (rr >> a.S_shift) : rr / a.S);
        const float *cc3 = a.new_xyz + (long long)bm * 3;
        // (a division like the loader this path replaces and the reference's `grouped_xyz /= radius`: the reciprocal form moved
        // the step's first loss from 30.08 to 32.06 — one ulp in three layers' offsets changes which votes the proposal FPS picks)
        *reinterpret_cast<uint2 *>(sA + lane * ldw + a.C) =
            pack4(make_float4((wx - cc3[0]) / a.radius, (wy - cc3[1]) / a.radius, (wz - cc3[2]) / a.radius, 0.f));
      }
    }
    // chunks in flight per lane: four, two for the wide BN-backward kernels (their raw operands — y, g or the pooled
    // triple — at four in flight pushed the kernel over 256 registers: one wave per SIMD)
    constexpr int UB = (LOADER == BNBWD && COUT >= 128) ? 2 : 4;
    for (int c0 = 0; !fastg && !wideg && !(PF && hoist_pf) && c0 < nch; c0 += (HOIST ? 64 * UB : 256)) {
      if (HOIST) {
        Raw8 raw[UB];
        const int kshift = __builtin_ctz(kc);  // kc | 64: a power of two (shifts instead of eight divisions per batch)
#pragma unroll
        for (int u = 0; u < UB; ++u) {  // unconditional (clamped) loads: branch-free, all in flight together
          const int c = min(c0 + 64 * u + lane, nch - 1);
          const int row = c >> kshift, ch = c & (kc - 1);
          raw_load8<LOADER>(a, row0 + row, ch * 8, sMeta[row], raw[u]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int c = c0 + 64 * u + lane;
          if (c < nch) {
            const int row = c >> kshift, ch = c & (kc - 1);
            *reinterpret_cast<uint4 *>(sA + row * ldw + ch * 8) = finish8<LOADER>(a, sMeta[row], raw[u], ca, cb, cc);
          }
        }
        continue;
      }
      float4 v0[4], v1[4];
      int vrow[4], vch[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // unconditional (clamped) loads: branch-free, all in flight together
        const int c = min(c0 + 64 * u + lane, nch - 1);
        // chunk -> (row, 16-byte column) by a multiply-shift (kc <= 36, c < 1152: exact): the division by a run-time kc
        // was a ~30-instruction sequence per chunk, 17 chunks per lane and tile at K = 272
        vrow[u] = (int)(((unsigned)c * kc_inv) >> 20);
        vch[u] = c - vrow[u] * kc;
        tile_chunk_load<LOADER>(a, row0 + vrow[u], vch[u] * 8, v0[u], v1[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 64 * u + lane;
        if (c < nch) *reinterpret_cast<uint4 *>(sA + vrow[u] * ldw + vch[u] * 8) = pack8(v0[u], v1[u]);
      }
    }
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = zero16();
    const bf16 *pa = sA + r * ldw + 8 * half;
    const bf16 *pw = sW + r * ldw + 8 * half;
    for (int g = 0; g < K / 16; ++g) {
      const bf16x8 av = *reinterpret_cast<const bf16x8 *>(pa + 16 * g);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(pw + (32 * ct) * ldw + 16 * g);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[ct], 0, 0, 0);
      }
    }

    // ---- epilogue.  Global traffic goes through the wave's LDS tile in 16-byte row chunks (a lane storing one
    // bf16 per instruction is store-issue bound); per-column reductions stay in the accumulator layout. ----
    if (EPI == STORE) {
      constexpr bool STAGED = COUT <= 128;  // wider outputs: the 2-byte LDS writes cost more than they save
      T *Y = reinterpret_cast<T *>(a.Yout) + (long long)row0 * a.ldout;
      float wr[16];  // multiplicity of this lane's 16 accumulator rows in the BatchNorm batch sums (1 without a row map)
#pragma unroll
      for (int i = 0; i < 16; ++i) wr[i] = __int_as_float(sMeta[acc_row(i, half)].y);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[ct][i];
          if (STAGED) sA[acc_row(i, half) * lde + 32 * ct + r] = __float2bfloat16(v);
          else st1(Y + (long long)acc_row(i, half) * a.ldout + 32 * ct + r, v);
          ps += wr[i] * v;
          pq += wr[i] * (v * v);
        }
        s1[ct] += (double)ps;
        s2[ct] += (double)pq;
      }
      if (STAGED)
        for (int c = lane; c < 32 * CC; c += 64) {
          const int row = c / CC, ch = c - row * CC;
          *reinterpret_cast<uint4 *>(Y + (long long)row * a.ldout + ch * 8) = *reinterpret_cast<const uint4 *>(sA + row * lde + ch * 8);
        }
    } else if (EPI == MASK) {
      const T *Yp = reinterpret_cast<const T *>(a.Yprev) + (long long)row0 * a.ldprev;
      for (int c = lane; c < 32 * CC; c += 64) {  // previous layer's pre-activation tile, coalesced
        const int row = c / CC, ch = c - row * CC;
        *reinterpret_cast<uint4 *>(sA + row * lde + ch * 8) = *reinterpret_cast<const uint4 *>(Yp + (long long)row * a.ldprev + ch * 8);
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 32 * ct + r;
        const float sc = a.p_scale[col], sh = a.p_shift[col], rs = a.p_rstd[col], nm = a.p_nmean_rstd[col];
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          bf16 *cell = sA + acc_row(i, half) * lde + col;
          const float y = __bfloat162float(*cell);
          const float g = (y * sc + sh > 0.f) ? acc[ct][i] : 0.f;
          *cell = __float2bfloat16(g);
          ps += g;
          pq += g * (y * rs + nm);
        }
        s1[ct] += (double)ps;
        s2[ct] += (double)pq;
      }
      T *G = reinterpret_cast<T *>(a.Yout) + (long long)row0 * a.ldout;
      for (int c = lane; c < 32 * CC; c += 64) {
        const int row = c / CC, ch = c - row * CC;
        *reinterpret_cast<uint4 *>(G + (long long)row * a.ldout + ch * 8) = *reinterpret_cast<const uint4 *>(sA + row * lde + ch * 8);
      }
    } else {
      scatter_rows<NCT>(a, acc, (int)(tile * 32), r, half);
    }
  }

  if (EPI == STORE || EPI == MASK) block_stats_to_slab<COUT>(s1, s2, (EPI == STORE) ? a.stats : a.tstats, r, half, wave);
  if (EPI == STORE || EPI == MASK) zero_unowned_slabs<COUT>((EPI == STORE) ? a.stats : a.tstats, a.R);
}

// ------------------------------------------------------------------------------------------------
// pool: out[bm][c] = relu(sel*scale + shift), sel = max_s Y (scale >= 0) or min_s Y (scale < 0);
// sel_idx[bm][c] = first s attaining it (the row the gradient is routed to, like max_pool2d).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(const T *__restrict__ Y, int ld, int S, int C, long long BM,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   float *__restrict__ out, unsigned char *__restrict__ sel_idx) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= BM * C) return;
  const long long bm = t / C;
  const int c = (int)(t - bm * C);
  const float sc = scale[c];
  float best = ld1(Y + (bm * S) * ld + c);
  int bi = 0;
  for (int s = 1; s < S; ++s) {
    const float v = ld1(Y + (bm * S + s) * ld + c);
    const bool better = sc >= 0.f ? (v > best) : (v < best);
    if (better) { best = v; bi = s; }
  }
  out[t] = fmaxf(0.f, best * sc + shift[c]);
  sel_idx[t] = (unsigned char)bi;
}

// Same result, 8 channels per thread: 16-byte (bf16) / 2 x 16-byte (fp32) row segments, four rows in flight.  The
// one-channel form reads 2 bytes per lane per dependent iteration and is latency bound (SA1: 174 us for 268 MB).
template <typename T>
__global__ __launch_bounds__(256) void pool8_kernel(const T *__restrict__ Y, int S_dense, int C, long long BM,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    float *__restrict__ out, unsigned char *__restrict__ sel_idx,
                                                    const int *__restrict__ rowptr, bf16 *__restrict__ out_bf) {
  const int c8n = C / 8;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= BM * c8n) return;
  const long long bm = t / c8n;
  const int c0 = (int)(t - bm * c8n) * 8;
  float sc[8], best[8];
  int bi[8];
  // compact row map: the ball's distinct rows are rowptr[bm] .. rowptr[bm+1]-1 (the padded copies never win a strict
  // comparison, so the selected position is the same as over the padded ball)
  const long long r0 = rowptr ? rowptr[bm] : bm * S_dense;
  const int S = rowptr ? rowptr[bm + 1] - (int)r0 : S_dense;
  const T *p = Y + r0 * C + c0;
  auto row8 = [&](const T *q, float (&v)[8]) {
    const float4 a = ld4(q), b = ld4(q + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  };
  row8(p, best);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[c0 + i];
    bi[i] = 0;
  }
  int s = 1;
  for (; s + 4 <= S; s += 4) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) row8(p + (long long)(s + u) * C, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool better = sc[i] >= 0.f ? (v[u][i] > best[i]) : (v[u][i] < best[i]);
        if (better) { best[i] = v[u][i]; bi[i] = s + u; }
      }
  }
  for (; s < S; ++s) {
    float v[8];
    row8(p + (long long)s * C, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool better = sc[i] >= 0.f ? (v[i] > best[i]) : (v[i] < best[i]);
      if (better) { best[i] = v[i]; bi[i] = s; }
    }
  }
  float *o = out + bm * C + c0;
  float4 o0, o1;
  o0.x = fmaxf(0.f, best[0] * sc[0] + shift[c0 + 0]); o0.y = fmaxf(0.f, best[1] * sc[1] + shift[c0 + 1]);
  o0.z = fmaxf(0.f, best[2] * sc[2] + shift[c0 + 2]); o0.w = fmaxf(0.f, best[3] * sc[3] + shift[c0 + 3]);
  o1.x = fmaxf(0.f, best[4] * sc[4] + shift[c0 + 4]); o1.y = fmaxf(0.f, best[5] * sc[5] + shift[c0 + 5]);
  o1.z = fmaxf(0.f, best[6] * sc[6] + shift[c0 + 6]); o1.w = fmaxf(0.f, best[7] * sc[7] + shift[c0 + 7]);
  *reinterpret_cast<float4 *>(o) = o0;
  *reinterpret_cast<float4 *>(o + 4) = o1;
  // the same row as bf16 (optional): what the NEXT level's gather layer rounds it to on its way into LDS — handed over so that
  // it can read 16-byte chunks of it instead (vlp3d_sa_pool_rows)
  if (out_bf != nullptr) *reinterpret_cast<uint4 *>(out_bf + bm * C + c0) = pack8(o0, o1);
  uint2 sb;
  sb.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
  sb.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
  *reinterpret_cast<uint2 *>(sel_idx + bm * C + c0) = sb;
}

// The same pooling with a WAVE per ball (bf16 storage, C in {128, 256}): Q = 64 / (C/8) lanes share a channel group and walk the
// ball's rows Q apart, then meet through shuffles (the better value; the smaller position on a tie: the reference's arg max is
// the first row that attains it).  A ball of the compact form has ~25 rows at SA1: the thread-per-(ball, channel group) form
// above walks them in seven dependent trips of four loads, this one in two (50 -> see DESIGN.md 4.20).
template <int C8N>
__global__ __launch_bounds__(256) void pool8_wave_kernel(const bf16 *__restrict__ Y, int S_dense, long long BM,
                                                         const float *__restrict__ scale, const float *__restrict__ shift,
                                                         float *__restrict__ out, unsigned char *__restrict__ sel_idx,
                                                         const int *__restrict__ rowptr, bf16 *__restrict__ out_bf) {
  constexpr int C = C8N * 8, Q = 64 / C8N;
  const int lane = threadIdx.x & 63;
  const long long bm = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bm >= BM) return;
  const int g = lane % C8N, q = lane / C8N, c0 = g * 8;
  const long long r0 = rowptr ? rowptr[bm] : bm * S_dense;
  const int S = rowptr ? rowptr[bm + 1] - (int)r0 : S_dense;
  const bf16 *p = Y + r0 * C + c0;
  float sc[8], best[8];
  int bi[8];
  auto row8 = [&](const bf16 *qp, float (&v)[8]) { unpack8(*reinterpret_cast<const uint4 *>(qp), v); };   // one 16-byte load
  const int s0 = min(q, S - 1);   // (a lane beyond the ball's rows repeats its last row: the tie rule discards it)
  row8(p + (long long)s0 * C, best);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[c0 + i];
    bi[i] = s0;
  }
  int s = q + Q;
  for (; s + 3 * Q < S; s += 4 * Q) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) row8(p + (long long)(s + u * Q) * C, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool better = sc[i] >= 0.f ? (v[u][i] > best[i]) : (v[u][i] < best[i]);
        if (better) { best[i] = v[u][i]; bi[i] = s + u * Q; }
      }
  }
  for (; s < S; s += Q) {
    float v[8];
    row8(p + (long long)s * C, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool better = sc[i] >= 0.f ? (v[i] > best[i]) : (v[i] < best[i]);
      if (better) { best[i] = v[i]; bi[i] = s; }
    }
  }
#pragma unroll
  for (int off = C8N; off < 64; off <<= 1) {   // the Q lanes of a channel group are C8N apart
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float ov = __shfl_xor(best[i], off);
      const int oi = __shfl_xor(bi[i], off);
      const bool better = sc[i] >= 0.f ? (ov > best[i]) : (ov < best[i]);
      if (better || (ov == best[i] && oi < bi[i])) { best[i] = ov; bi[i] = oi; }
    }
  }
  if (q != 0) return;
  float *o = out + bm * C + c0;
  float4 o0, o1;
  o0.x = fmaxf(0.f, best[0] * sc[0] + shift[c0 + 0]); o0.y = fmaxf(0.f, best[1] * sc[1] + shift[c0 + 1]);
  o0.z = fmaxf(0.f, best[2] * sc[2] + shift[c0 + 2]); o0.w = fmaxf(0.f, best[3] * sc[3] + shift[c0 + 3]);
  o1.x = fmaxf(0.f, best[4] * sc[4] + shift[c0 + 4]); o1.y = fmaxf(0.f, best[5] * sc[5] + shift[c0 + 5]);
  o1.z = fmaxf(0.f, best[6] * sc[6] + shift[c0 + 6]); o1.w = fmaxf(0.f, best[7] * sc[7] + shift[c0 + 7]);
  *reinterpret_cast<float4 *>(o) = o0;
  *reinterpret_cast<float4 *>(o + 4) = o1;
  if (out_bf != nullptr) *reinterpret_cast<uint4 *>(out_bf + bm * C + c0) = pack8(o0, o1);
  uint2 sb;
  sb.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
  sb.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
  *reinterpret_cast<uint2 *>(sel_idx + bm * C + c0) = sb;
}

// G3[(bm*S + s)][c] = (s == sel_idx[bm][c] && out[bm][c] > 0) ? dP[bm][c] : 0   (max-pool + ReLU backward)
template <typename T>
__global__ __launch_bounds__(256) void pool_grad_kernel(const float *__restrict__ dP, const float *__restrict__ out,
                                                        const unsigned char *__restrict__ sel_idx, int S, int C,
                                                        long long BM, T *__restrict__ G) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= BM * S * C) return;
  const long long row = t / C;
  const int c = (int)(t - row * C);
  const long long bm = row / S;
  const int s = (int)(row - bm * S);
  const long long o = bm * C + c;
  st1(G + t, (sel_idx[o] == s && out[o] > 0.f) ? dP[o] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[c][k] = sum_r dY[r][c] * A[r][k]   (COUT x K), fp32 MFMA, rows split over workgroups.
// Per 32-row tile the workgroup stages dY (32 x COUT, BN-backward applied) and A_{l-1} (32 x KP, gather or
// BN+ReLU applied) in LDS with the same vectorised loaders as row_gemm, then every wave owns a set of
// 32x32 output tiles: one MFMA step = 2 rows, A-operand lane (c = 32ct + r) reads dY[2kk+half][c], B-operand
// lane (k = 32kt + r) reads A[2kk+half][k] — both conflict-free 128-byte LDS rows.  One workgroup covers the
// WHOLE dW for its rows, so dY and A are read once.
// ------------------------------------------------------------------------------------------------
// Staging loaders of wgrad with the per-column constants hoisted: a thread always stages the same 4 columns of dY
// (256 % (COUT/4) == 0) and of a BN+ReLU A operand (256 % (K/4) == 0, host-checked).
template <typename T>
struct DyConsts {  // fp32: the five vectors, reference evaluation order; bf16: the folded form ca*g + (cb*y + cc)
  float4 k1, k2, k3, rs, nm;
  __device__ __forceinline__ void load(const RowGemmArgs &a, int col0) {
    k1 = ld4(a.k1 + col0); k2 = ld4(a.k2 + col0); k3 = ld4(a.k3 + col0);
    rs = ld4(a.rstd + col0); nm = ld4(a.nmean_rstd + col0);
    if (sizeof(T) == 2) {
      const float4 ca = k1;
      const float4 cb = make_float4(-(k1.x * k3.x) * rs.x, -(k1.y * k3.y) * rs.y, -(k1.z * k3.z) * rs.z, -(k1.w * k3.w) * rs.w);
      const float4 cc = make_float4(-k1.x * (k2.x + k3.x * nm.x), -k1.y * (k2.y + k3.y * nm.y), -k1.z * (k2.z + k3.z * nm.z),
                                    -k1.w * (k2.w + k3.w * nm.w));
      k1 = ca; k2 = cb; k3 = cc;
    }
  }
  __device__ __forceinline__ float one(float g, float y, float a, float b, float c, float r, float n) const {
    if (sizeof(T) == 2) return __builtin_fmaf(a, g, __builtin_fmaf(b, y, c));
    return a * (g - b - (y * r + n) * c);
  }
  __device__ __forceinline__ float4 apply(const float4 &g, const float4 &y) const {
    return make_float4(one(g.x, y.x, k1.x, k2.x, k3.x, rs.x, nm.x), one(g.y, y.y, k1.y, k2.y, k3.y, rs.y, nm.y),
                       one(g.z, y.z, k1.z, k2.z, k3.z, rs.z, nm.z), one(g.w, y.w, k1.w, k2.w, k3.w, rs.w, nm.w));
  }
};

template <typename T, int DYL>
__device__ __forceinline__ float4 load_dy4(const RowGemmArgs &a, int row, int col0, const DyConsts<T> &k) {
  const float4 y = ld4(reinterpret_cast<const T *>(a.Yin) + (long long)row * a.ldin + col0);
  if (DYL == PLAIN) return y;
  float4 g;
  if (a.pool_g != nullptr) {  // kernel-uniform
    const int bm = a.pool_shift >= 0 ? (row >> a.pool_shift) : (row / a.pool_S);
    const int sidx = row - bm * a.pool_S;
    const long long off = (long long)bm * a.ldin + col0;
    const float4 dp = ld4(a.pool_g + off);
    const uchar4 sl = *reinterpret_cast<const uchar4 *>(a.pool_sel + off);
    g = make_float4(sl.x == sidx ? dp.x : 0.f, sl.y == sidx ? dp.y : 0.f, sl.z == sidx ? dp.z : 0.f,
                    sl.w == sidx ? dp.w : 0.f);
  } else {
    g = ld4(reinterpret_cast<const T *>(a.Gin) + (long long)row * a.ldin + col0);
  }
  return k.apply(g, y);
}

// bf16 storage: 8 consecutive columns per staging element — ONE 16-byte load per operand (8-byte loads run at 0.54-0.70
// of the 16-byte rate, MI355X_MICROARCH.md), staged packed as bf16 in the LDS tile.
// Split loader of the weight-gradient kernel: `issue` only requests the operands (they stay in registers
// while the previous tile is contracted), `finish` does the arithmetic when the tile is written to LDS.  meta = words 1, 2
// of the row's compact-map entry ((ball << 8) | position in the ball, multiplicity), fetched ONE TILE FURTHER AHEAD, so
// that no request of a tile waits for another: the one-piece loader sat through up to eight dependent memory round trips
// per tile (map entry -> pooled gradient -> multiplicity, per staging element, each behind s_waitcnt vmcnt(0)) — 9.7 us per
// 32-row tile of SA1's last layer, 1.2 TB/s.
struct DyRaw8 {
  uint4 y, g;   // g: the gradient chunk, or the bits of dP[0..3] of the row's ball (pooled form)
  float4 dp1;   // pooled form: dP[4..7]
  uint2 sel;    // pooled form: the arg-max sample of the 8 channels
};

template <int DYL, bool POOL>
__device__ __forceinline__ void dy8_issue(const RowGemmArgs &a, int row, int col0, int2 meta, DyRaw8 &w) {
  w.y = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(a.Yin) + (long long)row * a.ldin + col0);
  if (DYL == PLAIN) return;
  if (POOL) {
    const long long off = (long long)(meta.x >> 8) * a.ldin + col0;
    const float4 d0 = ld4(a.pool_g + off);
    w.g = make_uint4(__float_as_uint(d0.x), __float_as_uint(d0.y), __float_as_uint(d0.z), __float_as_uint(d0.w));
    w.dp1 = ld4(a.pool_g + off + 4);
    w.sel = *reinterpret_cast<const uint2 *>(a.pool_sel + off);
  } else {
    w.g = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(a.Gin) + (long long)row * a.ldin + col0);
  }
}

// dY summed over the copies of a compact row = k1 G + w (cb y + cc); dense rows carry w = 1 in their (synthesised) meta
// word, for which the expression is the plain one bit for bit — no run-time branch left in the loader.
template <int DYL, bool POOL>
__device__ __forceinline__ uint4 dy8_finish(int2 meta, const DyRaw8 &w, const DyConsts<bf16> &klo, const DyConsts<bf16> &khi) {
  if (DYL == PLAIN) return w.y;
  float y[8], g[8];
  unpack8(w.y, y);
  if (POOL) {
    const int sidx = meta.x & 255;
    const float d0[4] = {__uint_as_float(w.g.x), __uint_as_float(w.g.y), __uint_as_float(w.g.z), __uint_as_float(w.g.w)};
    const float d1[4] = {w.dp1.x, w.dp1.y, w.dp1.z, w.dp1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g[i] = (int)((w.sel.x >> (8 * i)) & 255u) == sidx ? d0[i] : 0.f;
      g[4 + i] = (int)((w.sel.y >> (8 * i)) & 255u) == sidx ? d1[i] : 0.f;
    }
  } else {
    unpack8(w.g, g);
  }
  const float mult = __int_as_float(meta.y);
  float o[8];
  const float ca[8] = {klo.k1.x, klo.k1.y, klo.k1.z, klo.k1.w, khi.k1.x, khi.k1.y, khi.k1.z, khi.k1.w};
  const float cb[8] = {klo.k2.x, klo.k2.y, klo.k2.z, klo.k2.w, khi.k2.x, khi.k2.y, khi.k2.z, khi.k2.w};
  const float cc[8] = {klo.k3.x, klo.k3.y, klo.k3.z, klo.k3.w, khi.k3.x, khi.k3.y, khi.k3.z, khi.k3.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(ca[i], g[i], mult * __builtin_fmaf(cb[i], y[i], cc[i]));
  return pack8(make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7]));
}

struct WgradArgs {
  RowGemmArgs dy;   // BNBWD loader of this layer's dY (Gin, Yin, ldin = COUT, constants)
  RowGemmArgs src;  // loader of A_{l-1} (GATHER or BNRELU), K = src.K valid columns
  int KP;           // K rounded up to a multiple of 32
  float *partials;  // (gridDim.x x slab) fp32 scratch: one slab per workgroup, slab = COUT*K (+ COUT with colsum)
  long long tiles_per_block;
  int colsum;       // fp32 only: also reduce the columns of dY (the bias gradient of a linear layer) into slab[COUT*K..]
};

// BFM (fp32 storage only): stage the fp32 tiles as bf16 and contract with the bf16 MFMA — the timing configuration of the
// plain linear layers (same operand rounding as the bf16 grouped MLPs; accumulation stays fp32).
// The body takes its block coordinates as arguments: wgrad_kernel passes the launch's own, the batched linear form
// (rows_wgrad_batch_kernel below) the coordinates inside one job of its table.
// POOL (bf16 storage only): dY of the last layer is synthesised from the pooled tensors (w.dy.pool_g / pool_sel)
// RAWF (fp32 storage, no pooled gradient, no row map: the rows stacks and plain linear layers): like the bf16 loaders, the
// next tile's operands are only REQUESTED before the products and transformed when they are written to LDS.
// XB16 (RAWF, PLAIN operand): the A rows are bf16 in memory (an attention core's output, vlp3d_sdpa_fwd_io) — its own
// instantiation: a run-time test in the staging loop made hipcc branch around every load (117 -> 239 us for the batch).
template <typename T, int COUT, int LOADER, int MAXT, int DYL = BNBWD, bool BFM = false, bool POOL = false, bool RAWF = false, bool XB16 = false>  // MAXT = output tiles per wave
__device__ __forceinline__ void wgrad_body(const WgradArgs &w, const int bx, const int by, const int gx, const int gy) {
  extern __shared__ float lds[];
  constexpr int NCT = COUT / 32;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KP = w.KP, K = w.src.K, NKT = KP / 32, NT = NCT * NKT;
  constexpr bool BF = sizeof(T) == 2 || BFM;  // both tiles are staged as bf16 and contracted with the bf16 MFMA
  constexpr bool ST16 = sizeof(T) == 2;       // bf16 STORAGE: 8-column (16-byte) staging elements
  float *lds_dy = lds;               // fp32: [32][COUT]
  float *lds_a = lds + 32 * COUT;    // fp32: [32][KP]
  // bf16: [32][COUT + 4] and [32][KP + 4] shorts — row strides of 8 (mod 16) bytes put the two lane halves (rows +8)
  // on opposite bank halves for the column reads of the MFMA operands
  const int RSD = COUT + 4, RSA = KP + 4;
  short *ldb_dy = reinterpret_cast<short *>(lds);
  short *ldb_a = ldb_dy + 32 * RSD;

  f32x16 acc[MAXT];
  float csum = 0.f;
  int off_c[MAXT], off_k[MAXT];  // this wave's output tiles: t = wave + 4i -> (ct, kt), hoisted out of the hot loop
  bool tile_ok[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    acc[i] = zero16();
    const int t = wave + 4 * i;
    tile_ok[i] = t < NT;
    const int ct = tile_ok[i] ? t / NKT : 0;
    off_c[i] = 32 * ct;
    off_k[i] = 32 * (tile_ok[i] ? t - ct * NKT : 0);
  }

  // with a compact row map the number of rows is only known on the device: the (worst-case sized) grid re-divides them
  const bool compact = w.src.crow != nullptr;
  const long long ntiles = compact ? compact_tiles(w.src) : w.src.R / 32;
  const long long tpb = compact ? (ntiles + gx - 1) / gx : w.tiles_per_block;
  const long long t0 = min(ntiles, (long long)bx * tpb);
  const long long t1 = min(ntiles, t0 + tpb);
  // Staging is BRANCH-FREE: every global load is unconditional (clamped address) and only the LDS stores are
  // predicated — hipcc otherwise branches around each load and waits vmcnt(0) per element, serialising them.
  static_assert((32 * COUT / 4) % 256 == 0, "dY tile must split evenly over 256 threads");
  constexpr int NE_DY = 32 * COUT / 4 / 256;
  // bf16 storage: 8-column (16-byte) staging elements for dY and for a BN+ReLU / plain A operand (the gathered operand is
  // fp32: its 4-column elements already are 16 bytes)
  constexpr int NE_DY8 = (32 * COUT / 8 + 255) / 256;
  constexpr bool A8 = ST16 && LOADER != GATHER;
  // staging elements of the A tile per thread = 32*KP/4/256 = KP/32 = NKT <= 4*MAXT/NCT (MAXT >= NCT*NKT/4): sized by
  // the instantiation instead of the worst case 9 (five float4 of dead loads and registers at K = 128)
  constexpr int MAXE_A = (4 * MAXT / NCT) < 1 ? 1 : ((4 * MAXT / NCT) < 9 ? (4 * MAXT / NCT) : 9);
  const int KF = (LOADER == GATHER) ? w.src.C : K;       // columns read as plain 16-byte row slices
  const int kf4 = KF / 4, nef = 32 * kf4;
  // row of staging element e: K/4 is a power of two for the BN+ReLU / plain operands (host-checked: 256 % (K/4) == 0)
  const int kfs = (LOADER == GATHER) ? 0 : __builtin_ctz(kf4);
  auto row_of = [&](int e) -> int { return LOADER == GATHER ? e / kf4 : e >> kfs; };
  const int tc = (KP - KF) / 4, net = 32 * tc;           // tail: [xyz | zero padding] chunks, <= 256 elements
  float4 vdy[ST16 ? 1 : NE_DY], va[A8 ? 1 : MAXE_A], vt;
  float4 rgf[(RAWF && DYL == BNBWD) ? NE_DY : 1];  // RAWF: vdy holds the raw pre-activation, rgf the raw gradient
  constexpr int MAXE_A8 = (MAXE_A + 1) / 2;
  uint4 pa8[A8 ? MAXE_A8 : 1];         // bf16 storage: RAW operands of the next tile (arithmetic at the LDS store)
  // GATHER from bf16 feature rows (w.src.feat_bf, kernel-uniform): 8-column (16-byte) elements like the A8 operands — with 4-column
  // elements the loads shrink to 8 bytes each and the kernel, bound by its load instructions, got SLOWER (126 -> 176 us at SA1)
  constexpr bool G8 = ST16 && LOADER == GATHER;
  const bool gbf = G8 && w.src.feat_bf != nullptr;
  const int kg8 = (KF + 7) / 8, neg8 = 32 * kg8;          // 16-byte chunks per feature row (the last one may be half padding)
  const unsigned kg8_inv = ((1u << 20) + kg8 - 1) / max(kg8, 1);  // e / kg8 == (e * kg8_inv) >> 20 for e < 32 * kg8, kg8 <= 36
  uint4 pg8[G8 ? MAXE_A8 : 1];
  int pidx8[G8 ? MAXE_A8 : 1];
  DyRaw8 rdy[ST16 ? NE_DY8 : 1];
  int2 mc[ST16 ? NE_DY8 : 1], mn[ST16 ? NE_DY8 : 1];  // compact-map words of the rows in rdy / of the tile after it
  const int kf8 = KF / 8, nef8 = 32 * kf8, kfs8 = A8 ? __builtin_ctz(kf8 > 0 ? kf8 : 1) : 0;

  // column-block mode (gy > 1): this workgroup owns columns [coff, coff + COUT) of a wider dY — a 256-wide layer
  // runs as two 128-wide halves (64 instead of 144 accumulator registers: three waves per SIMD instead of one)
  const int coff = by * COUT;
  DyConsts<T> dyk, dyk2;
  if (DYL == BNBWD) {
    if (ST16) {
      dyk.load(w.dy, coff + (threadIdx.x % (COUT / 8)) * 8);
      dyk2.load(w.dy, coff + (threadIdx.x % (COUT / 8)) * 8 + 4);
    } else {
      dyk.load(w.dy, coff + (threadIdx.x % (COUT / 4)) * 4);
    }
  }
  float4 a_sc = make_float4(0.f, 0.f, 0.f, 0.f), a_sh = a_sc, a_sc2 = a_sc, a_sh2 = a_sc;
  if (LOADER == BNRELU) {
    const int k0 = A8 ? (threadIdx.x % kf8) * 8 : (threadIdx.x % kf4) * 4;
    a_sc = ld4(w.src.scale + k0);
    a_sh = ld4(w.src.shift + k0);
    if (A8) {
      a_sc2 = ld4(w.src.scale + k0 + 4);
      a_sh2 = ld4(w.src.shift + k0 + 4);
    }
  }
  // The gathered operand needs idx[row] before its feature row can be requested: two dependent memory latencies per
  // tile.  The indices therefore run ONE TILE FURTHER AHEAD than the data (pidx/tp hold the next tile's indices).
  int pidx[MAXE_A], tp = 0, tbm = 0;
  float tq[3] = {0.f, 0.f, 0.f}, tcn[3] = {0.f, 0.f, 0.f};  // raw [xyz | centre] of the tail chunk (arithmetic at the LDS store)
  auto fetch_idx = [&](long long tile) {
    if (LOADER != GATHER) return;
    const int row0 = (int)(tile * 32);
    if (gbf) {
      if constexpr (G8) {
#pragma unroll
        for (int j = 0; j < MAXE_A8; ++j) {
          const int e = min((int)threadIdx.x + 256 * j, neg8 - 1);
          const int row = (int)(((unsigned)e * kg8_inv) >> 20);
          pidx8[j] = compact ? w.src.crow[row0 + row].x : w.src.idx[row0 + row];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < MAXE_A; ++j) {
        const int e = min((int)threadIdx.x + 256 * j, nef - 1);
        pidx[j] = compact ? w.src.crow[row0 + row_of(e)].x : w.src.idx[row0 + row_of(e)];
      }
    }
    const int te = min((int)threadIdx.x, max(net, 1) - 1);
    const int trow = row0 + te / max(tc, 1);
    if (compact) {
      const int4 c = w.src.crow[trow];
      tp = c.x;
      tbm = c.y >> 8;
    } else {
      tp = w.src.idx[trow];
      tbm = w.src.S_shift >= 0 ? (trow >> w.src.S_shift) : trow / w.src.S;
    }
  };
  // M*S is a multiple of 32 (checked by the host), so a tile never straddles two scenes: the scene index is a
  // scalar that follows the (monotone) tile counter instead of two integer divisions per staged element.
  const long long tiles_per_scene = (LOADER == GATHER) ? ((long long)w.src.M * w.src.S) / 32 : 1;
  int scene = (LOADER == GATHER) ? (int)(t0 / tiles_per_scene) : 0;
  long long scene_end = (long long)(scene + 1) * tiles_per_scene;
  auto fetch = [&](long long tile) {
    const int row0 = (int)(tile * 32);
    if (LOADER == GATHER && !compact) {
      if (tile >= scene_end) {  // tiles advance by one and tiles_per_scene >= 1
        ++scene;
        scene_end += tiles_per_scene;
      }
    }
    const long long pbase = compact ? 0 : (long long)scene * w.src.N;  // compact rows carry global point rows
    if constexpr (ST16) {
#pragma unroll
      for (int j = 0; j < NE_DY8; ++j) {
        const int e = min((int)threadIdx.x + 256 * j, 32 * COUT / 8 - 1);
        dy8_issue<DYL, POOL>(w.dy, row0 + e / (COUT / 8), coff + (e % (COUT / 8)) * 8, mc[j], rdy[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE_DY; ++j) {
        const int e = threadIdx.x + 256 * j;
        if constexpr (RAWF) {
          const long long o = (long long)(row0 + e / (COUT / 4)) * w.dy.ldin + coff + (e % (COUT / 4)) * 4;
          vdy[j] = ld4(reinterpret_cast<const T *>(w.dy.Yin) + o);
          if (DYL == BNBWD) rgf[j] = ld4(reinterpret_cast<const T *>(w.dy.Gin) + o);
        } else {
          vdy[j] = load_dy4<T, DYL>(w.dy, row0 + e / (COUT / 4), coff + (e % (COUT / 4)) * 4, dyk);
        }
      }
    }
    if constexpr (A8) {
#pragma unroll
      for (int j = 0; j < MAXE_A8; ++j) {
        const int e = min((int)threadIdx.x + 256 * j, nef8 - 1);
        const int row = e >> kfs8, k0 = (e - (row << kfs8)) * 8;
        pa8[j] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(w.src.Yin) +
                                                  (long long)(row0 + row) * w.src.ldin + k0);
      }
    }
    if constexpr (G8) {
      if (gbf) {
#pragma unroll
        for (int j = 0; j < MAXE_A8; ++j) {
          const int e = min((int)threadIdx.x + 256 * j, neg8 - 1);
          const int row = (int)(((unsigned)e * kg8_inv) >> 20), k0 = (e - row * kg8) * 8;
          pg8[j] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16 *>(w.src.feat_bf) +
                                                    (pbase + pidx8[j]) * w.src.ldf + k0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < ((A8 || gbf) ? 0 : MAXE_A); ++j) {
      const int e = min((int)threadIdx.x + 256 * j, nef - 1);
      const int row = row_of(e), k0 = (e - row * kf4) * 4;
      const int rr = row0 + row;
      if (LOADER == GATHER) {
        va[j] = ld4(w.src.feat_pm + (pbase + pidx[j]) * w.src.C + k0);
      } else if (LOADER == BNRELU) {
        const float4 y = ld4(reinterpret_cast<const T *>(w.src.Yin) + (long long)rr * w.src.ldin + k0);
        if (RAWF) va[j] = y;
        else va[j] = make_float4(fmaxf(0.f, y.x * a_sc.x + a_sh.x), fmaxf(0.f, y.y * a_sc.y + a_sh.y),
                                 fmaxf(0.f, y.z * a_sc.z + a_sh.z), fmaxf(0.f, y.w * a_sc.w + a_sh.w));
      } else if (XB16) {  // bf16 rows: widened here, re-rounded (exactly) at the LDS store
        const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const short *>(w.src.Yin) + (long long)rr * w.src.ldin + k0);
        va[j] = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                            __uint_as_float(raw.y & 0xffff0000u));
      } else {
        va[j] = load_a4<T, LOADER>(w.src, rr, k0, 0, 0, 0);
      }
    }
    if (LOADER == GATHER) {
      const float *q = w.src.xyz + (pbase + tp) * 3;
      const float *c = w.src.new_xyz + (long long)tbm * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        tq[i] = q[i];
        tcn[i] = c[i];
      }
    }
  };
  auto tail_chunk = [&]() -> float4 {  // [dx, dy, dz, 0] / radius for chunk 0 of the tail, zero padding after it
    if (LOADER != GATHER) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int te = min((int)threadIdx.x, max(net, 1) - 1);
    const bool first = (te % max(tc, 1)) == 0;
    const float x = (tq[0] - tcn[0]) / w.src.radius, y = (tq[1] - tcn[1]) / w.src.radius, z = (tq[2] - tcn[2]) / w.src.radius;
    return make_float4(first ? x : 0.f, first ? y : 0.f, first ? z : 0.f, 0.f);
  };

  auto fetch_meta = [&](long long tile) {  // -> mn: ((ball << 8) | position in the ball, multiplicity) of the staged rows
    if constexpr (ST16) {
      const int row0 = (int)(tile * 32);
#pragma unroll
      for (int j = 0; j < NE_DY8; ++j) {
        const int row = row0 + min((int)threadIdx.x + 256 * j, 32 * COUT / 8 - 1) / (COUT / 8);
        if (DYL == BNBWD && w.dy.crow) {  // kernel-uniform
          const int4 c = w.dy.crow[row];
          mn[j] = make_int2(c.y, c.z);
        } else if (POOL) {  // dense rows: (ball, sample) from the row number, multiplicity 1
          const int bm = w.dy.pool_shift >= 0 ? (row >> w.dy.pool_shift) : (row / w.dy.pool_S);
          mn[j] = make_int2((bm << 8) | (row - bm * w.dy.pool_S), __float_as_int(1.f));
        } else {
          mn[j] = make_int2(0, __float_as_int(1.f));
        }
      }
    }
  };
  auto take_meta = [&]() {
    if constexpr (ST16) {
#pragma unroll
      for (int j = 0; j < NE_DY8; ++j) mc[j] = mn[j];
    }
  };
  if (t0 < t1) {
    fetch_idx(t0);
    fetch_meta(t0);
    take_meta();
    fetch(t0);
    fetch_idx(min(t0 + 1, t1 - 1));
    fetch_meta(min(t0 + 1, t1 - 1));
  }
  for (long long tile = t0; tile < t1; ++tile) {
    __syncthreads();  // the previous tile's MFMA reads are done
    vt = tail_chunk();
    if constexpr (RAWF && !ST16) {
      if (DYL == BNBWD) {
#pragma unroll
        for (int j = 0; j < NE_DY; ++j) vdy[j] = dyk.apply(rgf[j], vdy[j]);
      }
      if (LOADER == BNRELU) {
#pragma unroll
        for (int j = 0; j < (A8 ? 0 : MAXE_A); ++j)
          va[j] = make_float4(fmaxf(0.f, va[j].x * a_sc.x + a_sh.x), fmaxf(0.f, va[j].y * a_sc.y + a_sh.y),
                              fmaxf(0.f, va[j].z * a_sc.z + a_sh.z), fmaxf(0.f, va[j].w * a_sc.w + a_sh.w));
      }
    }
    if (BF) {
      // rows of the LDS tiles are 8 (mod 16) bytes apart (bank layout of the operand reads): two 8-byte stores per element
      if constexpr (ST16) {
#pragma unroll
        for (int j = 0; j < NE_DY8; ++j) {
          const int e = threadIdx.x + 256 * j;
          if (e < 32 * COUT / 8) {
            const uint4 v = dy8_finish<DYL, POOL>(mc[j], rdy[j], dyk, dyk2);
            short *q = ldb_dy + (e / (COUT / 8)) * RSD + (e % (COUT / 8)) * 8;
            *reinterpret_cast<uint2 *>(q) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2 *>(q + 4) = make_uint2(v.z, v.w);
          }
        }
      } else {  // fp32 storage, bf16 MFMA: 4-column (16-byte) elements rounded here
#pragma unroll
        for (int j = 0; j < NE_DY; ++j) {
          const int e = threadIdx.x + 256 * j;
          *reinterpret_cast<uint2 *>(ldb_dy + (e / (COUT / 4)) * RSD + (e % (COUT / 4)) * 4) = pack4(vdy[j]);
        }
      }
      if constexpr (A8) {
#pragma unroll
        for (int j = 0; j < MAXE_A8; ++j) {
          const int e = threadIdx.x + 256 * j;
          if (e < nef8) {
            const int row = e >> kfs8;
            uint4 v = pa8[j];
            if (LOADER == BNRELU) {
              float y[8];
              unpack8(v, y);
              v = pack8(make_float4(fmaxf(0.f, y[0] * a_sc.x + a_sh.x), fmaxf(0.f, y[1] * a_sc.y + a_sh.y),
                                    fmaxf(0.f, y[2] * a_sc.z + a_sh.z), fmaxf(0.f, y[3] * a_sc.w + a_sh.w)),
                        make_float4(fmaxf(0.f, y[4] * a_sc2.x + a_sh2.x), fmaxf(0.f, y[5] * a_sc2.y + a_sh2.y),
                                    fmaxf(0.f, y[6] * a_sc2.z + a_sh2.z), fmaxf(0.f, y[7] * a_sc2.w + a_sh2.w)));
            }
            short *q = ldb_a + row * RSA + (e - (row << kfs8)) * 8;
            *reinterpret_cast<uint2 *>(q) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2 *>(q + 4) = make_uint2(v.z, v.w);
          }
        }
      }
      if constexpr (G8) {
        if (gbf) {
#pragma unroll
          for (int j = 0; j < MAXE_A8; ++j) {
            const int e = threadIdx.x + 256 * j;
            if (e < neg8) {
              const int row = (int)(((unsigned)e * kg8_inv) >> 20), k0 = (e - row * kg8) * 8;
              short *q = ldb_a + row * RSA + k0;
              *reinterpret_cast<uint2 *>(q) = make_uint2(pg8[j].x, pg8[j].y);
              // the second half of a row's last chunk is the tail's [dx, dy, dz, 0] (written below by another thread)
              if (k0 + 4 < KF) *reinterpret_cast<uint2 *>(q + 4) = make_uint2(pg8[j].z, pg8[j].w);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < ((A8 || gbf) ? 0 : MAXE_A); ++j) {
        const int e = threadIdx.x + 256 * j;
        if (e < nef) {
          const int row = row_of(e);
          *reinterpret_cast<uint2 *>(ldb_a + row * RSA + (e - row * kf4) * 4) = pack4(va[j]);
        }
      }
      if ((int)threadIdx.x < net) {
        const int row = threadIdx.x / tc;
        *reinterpret_cast<uint2 *>(ldb_a + row * RSA + KF + (threadIdx.x - row * tc) * 4) = pack4(vt);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE_DY; ++j) *reinterpret_cast<float4 *>(lds_dy + 4 * (threadIdx.x + 256 * j)) = vdy[j];
#pragma unroll
      for (int j = 0; j < MAXE_A; ++j) {
        const int e = threadIdx.x + 256 * j;
        if (e < nef) {
          const int row = row_of(e);
          *reinterpret_cast<float4 *>(lds_a + row * KP + (e - row * kf4) * 4) = va[j];
        }
      }
      if ((int)threadIdx.x < net) {
        const int row = threadIdx.x / tc;
        *reinterpret_cast<float4 *>(lds_a + row * KP + KF + (threadIdx.x - row * tc) * 4) = vt;
      }
    }
    __syncthreads();
    if (tile + 1 < t1) {
      take_meta();
      fetch(tile + 1);
      fetch_idx(min(tile + 2, t1 - 1));
      fetch_meta(min(tile + 2, t1 - 1));
    }
    if (!ST16 && w.colsum && (int)threadIdx.x < COUT) {
      if (BFM) {  // the staged tile is bf16: the bias gradient sums the rounded values
#pragma unroll 8
        for (int row = 0; row < 32; ++row)
          csum += __uint_as_float(((unsigned)(unsigned short)ldb_dy[row * RSD + threadIdx.x]) << 16);
      } else {
#pragma unroll 8
        for (int row = 0; row < 32; ++row) csum += lds_dy[row * COUT + threadIdx.x];
      }
    }
    if (BF) {
      // contraction over the 32 rows in two 16-row steps; lane (col, half) gathers its column's 8 rows
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const short *pa = ldb_dy + (16 * kb + 8 * half) * RSD + r;
        const short *pb = ldb_a + (16 * kb + 8 * half) * RSA + r;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          bf16x8 av, bv;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            av[j] = pa[j * RSD + off_c[i]];
            bv[j] = pb[j * RSA + off_k[i]];
          }
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
        }
      }
      continue;
    }
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const float *pa = lds_dy + (2 * kk + half) * COUT + r;
      const float *pb = lds_a + (2 * kk + half) * KP + r;
#pragma unroll
      for (int i = 0; i < MAXT; ++i)  // a wave with fewer than MAXT tiles recomputes tile (0,0) into an unused accumulator:
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[off_c[i]], pb[off_k[i]], acc[i], 0, 0, 0);  // no branches
    }
  }
  // partial dW of this workgroup's rows: plain coalesced stores into its own slab (summed by wgrad_reduce);
  // thousands of workgroups atomically adding into the same 36 KB matrix run an order of magnitude slower
  // slab of this row chunk: [Ntot x K | Ntot bias sums (colsum)] with Ntot = COUT * gy; this workgroup fills
  // rows coff .. coff + COUT of the matrix and its part of the bias segment
  const long long ntot = (long long)COUT * gy;
  float *slab0 = w.partials + (long long)bx * (ntot * K + (w.colsum ? ntot : 0));
  float *slab = slab0 + (long long)coff * K;
  if (!ST16 && w.colsum && (int)threadIdx.x < COUT) slab0[ntot * K + coff + threadIdx.x] = csum;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + 4 * i;
    if (t >= NT) continue;
    const int ct = t / NKT, kt = t - ct * NKT;
    const int k = 32 * kt + r;
    if (k >= K) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) slab[(long long)(32 * ct + acc_row(e, half)) * K + k] = acc[i][e];
  }
}

template <typename T, int COUT, int LOADER, int MAXT, int DYL = BNBWD, bool BFM = false, bool POOL = false>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs w) {
  wgrad_body<T, COUT, LOADER, MAXT, DYL, BFM, POOL>(w, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// Weight gradients of SEVERAL plain linear layers / rows-stack layers in one launch (bf16-MFMA timing configuration,
// 64-column blocks): the backward pass of the grounding step issues 33 + 13 such launches of 6-19 us each, every one of them
// a few hundred workgroups that leave most of the chip idle behind a memory latency; none of them feeds anything but the
// optimiser.  The step driver therefore queues them (operands stay alive) and runs them together: blockIdx.z = job,
// blockIdx.x / y = the job's row group / column block (workgroups beyond a job's own grid leave at once).
struct RowsWgradJob {  // vlp3d_rows_wgrad's operands (a plain linear layer: Ypre = bn5 = a_scale = a_shift = NULL)
  const float *G, *Ypre, *bn5, *X, *a_scale, *a_shift;
  float *partials;
  int R, K, N, ldg, lda, nblk, tpb, colsum;
};
constexpr int ROWS_WGRAD_BATCH = 40;  // 40 x 88 bytes of kernel arguments
struct RowsWgradBatch {
  RowsWgradJob j[ROWS_WGRAD_BATCH];
};
template <int LOADER, int DYL, int MAXT, bool XB16 = false>
__global__ __launch_bounds__(256) void rows_wgrad_batch_kernel(RowsWgradBatch t) {
  const RowsWgradJob &jb = t.j[blockIdx.z];
  const int ncb = jb.N / 64;
  if ((int)blockIdx.x >= jb.nblk || (int)blockIdx.y >= ncb) return;
  WgradArgs w = {};
  w.colsum = jb.colsum;
  w.src.K = jb.K; w.src.R = jb.R; w.src.Yin = jb.X; w.src.ldin = jb.lda; w.src.scale = jb.a_scale; w.src.shift = jb.a_shift;
  w.dy.ldin = jb.ldg;
  if (DYL == BNBWD) {
    const int N = jb.N;
    w.dy.Gin = jb.G; w.dy.Yin = jb.Ypre;
    w.dy.rstd = jb.bn5; w.dy.nmean_rstd = jb.bn5 + N; w.dy.k1 = jb.bn5 + 2 * N; w.dy.k2 = jb.bn5 + 3 * N;
    w.dy.k3 = jb.bn5 + 4 * N;
  } else {
    w.dy.Yin = jb.G;
  }
  w.KP = (jb.K + 31) & ~31;
  w.partials = jb.partials;
  w.tiles_per_block = jb.tpb;
  wgrad_body<float, 64, LOADER, MAXT, DYL, true, false, true, XB16>(w, blockIdx.x, blockIdx.y, jb.nblk, ncb);
}

// dW[i] = sum_b partials[b][i]: a block sums 64 consecutive elements — 16 threads x float4 — in 16 slab-groups
// (each thread nblk/16 independent 16-byte loads), then folds the groups through LDS.  n % 4 == 0.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partials, int nblk, int n,
                                                           float *__restrict__ dW) {
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
    *reinterpret_cast<float4 *>(dW + i) = t;
  }
}

// Same sum, but the (N x K) matrix part goes to dW with row stride ldo (a K-slice of a wider weight gradient) and the
// optional trailing N bias sums go to dbias.  One thread per output element, 16 slab groups like wgrad_reduce.
__global__ __launch_bounds__(256) void wgrad_reduce_strided_kernel(const float *__restrict__ partials, int nblk, int N, int K,
                                                                   int with_bias, float *__restrict__ dW, int ldo,
                                                                   float *__restrict__ dbias) {
  __shared__ float red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int n = N * K + (with_bias ? N : 0);
  const int i = blockIdx.x * 16 + q;
  float s = 0.f;
  if (i < n) {
#pragma unroll 4
    for (int b = grp; b < nblk; b += 16) s += partials[(long long)b * n + i];
  }
  red[grp][q] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = red[0][q];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g][q];
    if (i < N * K) dW[(long long)(i / K) * ldo + (i % K)] = t;
    else dbias[i - N * K] = t;
  }
}

// MANY slab sums in ONE launch (vlp3d_slab_reduce_batch): the weight-gradient kernels of a backward pass leave their
// per-workgroup slabs behind (defer_reduce != 0) and the step driver sums all of them at the end of backward — one launch
// instead of one per layer (51 launches of 5-9 us each at cfg2).  A block sums 64 consecutive elements of one entry, as
// wgrad_reduce; the entry table travels by value in the kernel arguments.
constexpr int SLAB_BATCH = 40;
struct SlabBatch {
  vlp3d_slab_reduce_desc d[SLAB_BATCH];
  int first_block[SLAB_BATCH + 1];
  int count;
};
__global__ __launch_bounds__(256) void slab_reduce_batch_kernel(SlabBatch t) {
  __shared__ float4 red[16][16];
  int e = 0;
  while (e + 1 < t.count && (int)blockIdx.x >= t.first_block[e + 1]) ++e;  // block-uniform
  const vlp3d_slab_reduce_desc &d = t.d[e];
  const int n = d.n_mat + d.n_bias;
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = ((int)blockIdx.x - t.first_block[e]) * 64 + 4 * q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
#pragma unroll 4
    for (int b = grp; b < d.nblk; b += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(d.partials + (long long)b * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[grp][q] = s;
  __syncthreads();
  if (grp != 0 || i >= n) return;
  float4 a = red[0][q];
#pragma unroll
  for (int g = 1; g < 16; ++g) {
    const float4 v = red[g][q];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const float v4[4] = {a.x, a.y, a.z, a.w};
  if (i >= d.n_mat) {  // n_mat % 4 == 0: the four elements are on the same side
#pragma unroll
    for (int j = 0; j < 4; ++j) d.dbias[i - d.n_mat + j] = v4[j];
    return;
  }
  const int row = i / d.K, k = i - row * d.K;  // K % 4 == 0: same row
  float *o = d.dst + (long long)row * d.ldo;
  if (d.ncol_out > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k + j < d.ncol_out) o[(k + j + d.rot) % d.ncol_out] = v4[j];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[k + j] = v4[j];
  }
}

// ------------------------------------------------------------------------------------------------
// Per-channel bookkeeping of the fused layer (each replaces ~20 tiny framework kernels per BatchNorm):
// ------------------------------------------------------------------------------------------------
// Sum the per-workgroup slabs [nslab][2][C] for SUMC columns per block: 256 threads = 64 slab-groups x 4 columns (a
// thread adds nslab/64 slabs: with 16 groups the 64 dependent fp64 loads per thread made these kernels 15 us each).
// Returns (sum of [0][c], sum of [1][c]) to the threads of group 0 (threadIdx.x < SUMC); c = blockIdx.x*SUMC + col.
constexpr int SUMC = 4;
__device__ __forceinline__ void slab_sum16(const double *__restrict__ slabs, int nslab, int C, double &s0, double &s1) {
  __shared__ double red[2][4][SUMC];
  const int col = threadIdx.x & (SUMC - 1), grp = threadIdx.x / SUMC;
  const int c = blockIdx.x * SUMC + col;
  double a = 0.0, b = 0.0;
  if (c < C) {
    // all of a thread's slabs requested before the first add (<= 16 per thread at the capped grids: one memory round
    // trip instead of four; the adds keep their order, so the sums keep their bits)
    for (int k0 = grp; k0 < nslab; k0 += 64 * 16) {
      double va[16], vb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {  // unconditional loads at clamped addresses (a test per load makes hipcc branch around it)
        const int k = min(k0 + 64 * u, nslab - 1);
        va[u] = slabs[(size_t)k * 2 * C + c];
        vb[u] = slabs[(size_t)k * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + 64 * u < nslab) {
          a += va[u];
          b += vb[u];
        }
    }
  }
  // the 16 groups of a wave meet through shuffles (lanes 4 apart share a column), the four waves through LDS: one barrier
  // instead of the seven of a 64-leaf tree in LDS; a fixed order either way
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < SUMC) {
    red[0][wave][col] = a;
    red[1][wave][col] = b;
  }
  __syncthreads();
  s0 = (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]);
  s1 = (red[1][0][col] + red[1][1][col]) + (red[1][2][col] + red[1][3][col]);
}

// bn_fold: batch statistics (fp64 sums) -> vec[4][C] = [scale | shift | rstd | -mean*rstd], running-stat update.
__global__ void bn_fold_kernel(const double *__restrict__ stats, int nslab, const float *__restrict__ gamma,
                               const float *__restrict__ beta, float *__restrict__ running_mean,
                               float *__restrict__ running_var, int C, double R, float eps, float momentum,
                               int training, float *__restrict__ vec, const float *__restrict__ mean_shift) {
  double s = 0.0, q = 0.0;
  // the per-channel operands of the tail are requested BEFORE the slab sum (they used to cost a second memory round trip behind it)
  const int c = blockIdx.x * SUMC + (threadIdx.x & (SUMC - 1));
  const bool tail = threadIdx.x < SUMC && c < C;
  const int cc = min(c, C - 1);
  const float g_c = gamma[cc], b_c = beta[cc];
  const float rm_c = running_mean != nullptr ? running_mean[cc] : 0.f, rv_c = running_var != nullptr ? running_var[cc] : 0.f;
  const float ms_c = mean_shift != nullptr ? mean_shift[cc] : 0.f;
  if (training) slab_sum16(stats, nslab, C, s, q);
  if (!tail) return;
  double mean, var;
  if (training) {
    mean = s / R;
    var = q / R - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean != nullptr) {  // nn.BatchNorm: unbiased variance in the running estimate
      // mean_shift: a conv bias in front of the BatchNorm shifts the batch mean the running estimate tracks (and nothing else)
      running_mean[c] = (1.f - momentum) * rm_c + momentum * ((float)mean + ms_c);
      running_var[c] = (1.f - momentum) * rv_c + momentum * (float)(var * (R / (R > 1.0 ? R - 1.0 : 1.0)));
    }
  } else {
    mean = rm_c;
    var = rv_c;
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double scale = (double)g_c * rstd;
  vec[c] = (float)scale;
  vec[C + c] = (float)((double)b_c - mean * scale);
  vec[2 * C + c] = (float)rstd;
  vec[3 * C + c] = (float)(-mean * rstd);
}

// bn5: backward constants [rstd | -mean*rstd | gamma*rstd | mean(g) | mean(g*yhat)] + d gamma, d beta
__global__ void bn5_kernel(const float *__restrict__ vec, const float *__restrict__ gamma,
                           const double *__restrict__ tslabs, int nslab, int C, double R, int training,
                           float *__restrict__ bn5, float *__restrict__ dgamma, float *__restrict__ dbeta) {
  double t[2];
  const int c = blockIdx.x * SUMC + (threadIdx.x & (SUMC - 1));
  const int cc = min(c, C - 1);
  const float rstd = vec[2 * C + cc], nmr = vec[3 * C + cc], g_c = gamma[cc];  // requested before the slab sum, not behind it
  slab_sum16(tslabs, nslab, C, t[0], t[1]);
  if (threadIdx.x >= SUMC || c >= C) return;
  bn5[c] = rstd;
  bn5[C + c] = nmr;
  bn5[2 * C + c] = g_c * rstd;
  bn5[3 * C + c] = training ? (float)(t[0] / R) : 0.f;
  bn5[4 * C + c] = training ? (float)(t[1] / R) : 0.f;
  dbeta[c] = (float)t[0];
  dgamma[c] = (float)t[1];
}

// BN-backward reductions of the LAST layer straight from the pooled tensors: only the selected sample of each
// ball carries gradient, and there z = gamma*yhat + beta = out, so yhat = (out - beta)/gamma.
__global__ __launch_bounds__(256) void pool_tstats_kernel(const float *__restrict__ dP, const float *__restrict__ out,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, long long BM, int C,
                                                          long long rows_per_block, double *__restrict__ t,
                                                          float *__restrict__ gsel) {
  // A thread owns a channel and walks rows.  With <= 128 channels in the block the two halves of the workgroup take the two
  // halves of the block's rows (the idle half used to watch 128 dependent rows go by: 15.8 us at SA1) and meet in LDS; sixteen
  // rows are in flight instead of eight.
  __shared__ double sh[2][128];
  const int cbase = blockIdx.x * 256;
  const int cw = min(C - cbase, 256);
  const bool split = cw <= 128;
  const int tc = split ? ((int)threadIdx.x & 127) : (int)threadIdx.x;
  const int part = split ? ((int)threadIdx.x >> 7) : 0;
  const int c = cbase + tc;
  const bool active = tc < cw;
  const long long b0 = (long long)blockIdx.y * rows_per_block, b1 = min(BM, b0 + rows_per_block);
  const long long mid = split ? b0 + (b1 - b0 + 1) / 2 : b1;
  const long long r0 = part == 0 ? b0 : mid, r1 = part == 0 ? mid : b1;
  double s1 = 0.0, s2 = 0.0;
  if (active) {
    const float g = gamma[c], b = beta[c];
    const float inv = g == 0.f ? 0.f : 1.f / g;
    long long r = r0;
    for (; r + 16 <= r1; r += 16) {  // sixteen rows in flight: the one-row loop was a chain of dependent load latencies
      float o[16], dp[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        o[u] = out[(r + u) * C + c];
        dp[u] = dP[(r + u) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float d = o[u] > 0.f ? dp[u] : 0.f;
        gsel[(r + u) * C + c] = d;
        s1 += d;
        s2 += d * ((o[u] - b) * inv);
      }
    }
    for (; r < r1; ++r) {
      const float o = out[r * C + c];
      const float d = o > 0.f ? dP[r * C + c] : 0.f;
      gsel[r * C + c] = d;
      s1 += d;
      s2 += d * ((o - b) * inv);
    }
  }
  if (split) {  // (block-uniform)
    if (part == 1 && active) {
      sh[0][tc] = s1;
      sh[1][tc] = s2;
    }
    __syncthreads();
    if (part == 0 && active) {
      s1 += sh[0][tc];
      s2 += sh[1][tc];
    }
  }
  if (part == 0 && active) {
    t[((size_t)blockIdx.y * 2 + 0) * C + c] = s1;
    t[((size_t)blockIdx.y * 2 + 1) * C + c] = s2;
  }
}

void set_pool(RowGemmArgs &a, const float *pool_g, const unsigned char *pool_sel, int pool_S) {
  a.pool_g = pool_g;
  a.pool_sel = pool_sel;
  a.pool_S = pool_S;
  a.pool_shift = -1;
  for (int sh = 0; sh < 31; ++sh)
    if ((1 << sh) == pool_S) a.pool_shift = sh;
}

unsigned grid_tiles(long long R) { return stat_slabs_of(R); }

template <typename T, int LOADER, int EPI>
int launch_row_gemm_t(int cout, const RowGemmArgs &a, hipStream_t s) {
  const dim3 grid(grid_tiles(a.R)), block(256);
  switch (cout) {
    case 32: hipLaunchKernelGGL((row_gemm_kernel<T, 32, LOADER, EPI>), grid, block, 0, s, a); break;
    case 64: hipLaunchKernelGGL((row_gemm_kernel<T, 64, LOADER, EPI>), grid, block, 0, s, a); break;
    case 128: hipLaunchKernelGGL((row_gemm_kernel<T, 128, LOADER, EPI>), grid, block, 0, s, a); break;
    case 160: hipLaunchKernelGGL((row_gemm_kernel<T, 160, LOADER, EPI>), grid, block, 0, s, a); break;
    case 256: hipLaunchKernelGGL((row_gemm_kernel<T, 256, LOADER, EPI>), grid, block, 0, s, a); break;
    case 288: hipLaunchKernelGGL((row_gemm_kernel<T, 288, LOADER, EPI>), grid, block, 0, s, a); break;
    default: return VLP3D_EINVAL;
  }
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

template <int COUT, int LOADER, int EPI>
int launch_lds_c(const RowGemmArgs &a, hipStream_t s) {
  const size_t wtile = 32 * (size_t)((a.K > COUT ? a.K : COUT) + 8);  // A operand / epilogue staging, per wave
  const size_t lds = ((size_t)COUT * (a.K + 8) + 4 * wtile) * sizeof(bf16) + 4 * 32 * sizeof(int2);  // + row-map words
  const size_t lds_static = (EPI == SCATTER) ? 0 : sizeof(double) * 4 * 2 * COUT;  // block_stats_to_slab
  if (lds + lds_static > 160 * 1024) return VLP3D_EINVAL;
  auto kern = row_gemm_lds_kernel<COUT, LOADER, EPI>;
  bool wide = false;
  if constexpr (LOADER == GATHER && EPI == STORE && COUT == 128) {
    // the 256-channel levels (K = 272): all row-map words, then all feature rows of a tile in flight at once
    const int kc = a.K / 8;
    static const bool wide_on = !(getenv("VLP3D_SA_WIDE") && atoi(getenv("VLP3D_SA_WIDE")) == 0);
    if (wide_on && 32 * kc > 64 * 10 && 32 * kc <= 64 * 18 && kc <= 36 && a.C % 8 == 0 && (a.tile_scene || a.crow != nullptr)) {
      kern = row_gemm_lds_kernel<COUT, LOADER, EPI, true>;
      wide = true;
    }
  }
  // bf16 feature rows are read by the gather fast path (32 K/8 <= 640 chunks per tile) and by the WIDE instantiation only
  if (LOADER == GATHER && a.feat_bf != nullptr && !wide && 32 * (a.K / 8) > 64 * 10) return VLP3D_EINVAL;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  // more than half a CU's LDS: ONE workgroup per CU is resident, and a second wave of workgroups would stage the weight again
  // behind the first (SA3's gather layer, K = 272: 512 workgroups of one tile per wave took two rounds of ~25 us) — cap the grid
  // at the CU count and let the waves stride over the tiles instead
  unsigned grid = grid_tiles(a.R);
  static const bool cap_on = !(getenv("VLP3D_SA_GRIDCAP") && atoi(getenv("VLP3D_SA_GRIDCAP")) == 0);
  if (cap_on && lds + lds_static > 80 * 1024 && grid > 256) grid = 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

template <int LOADER, int EPI>
int launch_row_gemm_lds(int cout, const RowGemmArgs &a, hipStream_t s) {
  switch (cout) {
    case 64: return launch_lds_c<64, LOADER, EPI>(a, s);
    case 128: return launch_lds_c<128, LOADER, EPI>(a, s);
    case 160: return launch_lds_c<160, LOADER, EPI>(a, s);
    case 256: return launch_lds_c<256, LOADER, EPI>(a, s);
    case 288: return launch_lds_c<288, LOADER, EPI>(a, s);
    default: return VLP3D_EINVAL;
  }
}

template <int LOADER, int EPI>
int launch_bf16(int cout, const RowGemmArgs &a, hipStream_t s) {
  // the LDS form needs K % 16 == 0 (always true for bf16 storage here); fall back to the direct form otherwise
  const bool lds_shape = cout == 64 || cout == 128 || cout == 160 || cout == 256 || cout == 288;
  const size_t need = ((size_t)cout * (a.K + 8) + 128 * (size_t)((a.K > cout ? a.K : cout) + 8)) * 2 + 64 * (size_t)cout;
  const bool cols_fixed = LOADER == GATHER || (a.K % 8 == 0 && 64 % (a.K / 8) == 0);  // hoisted per-lane constants
  if (lds_shape && a.K % 16 == 0 && cols_fixed && need <= 158 * 1024)
    return launch_row_gemm_lds<LOADER, EPI>(cout, a, s);
  // the compact row map is implemented by the LDS kernels, and by the direct kernel for the wide scatter layer only
  if (a.crow && !(LOADER == BNBWD && EPI == SCATTER)) return VLP3D_EINVAL;
  return launch_row_gemm_t<bf16, LOADER, EPI>(cout, a, s);
}

template <typename T>
int launch_row_gemm(int loader, int epi, int cout, const RowGemmArgs &a, hipStream_t s) {
  if (sizeof(T) == 2) {
    if (loader == GATHER && epi == STORE) return launch_bf16<GATHER, STORE>(cout, a, s);
    if (loader == BNRELU && epi == STORE) return launch_bf16<BNRELU, STORE>(cout, a, s);
    if (loader == BNBWD && epi == MASK) return launch_bf16<BNBWD, MASK>(cout, a, s);
    if (loader == BNBWD && epi == SCATTER) return launch_bf16<BNBWD, SCATTER>(cout, a, s);
    return VLP3D_EINVAL;
  }
  if (loader == GATHER && epi == STORE) return launch_row_gemm_t<T, GATHER, STORE>(cout, a, s);
  if (loader == BNRELU && epi == STORE) return launch_row_gemm_t<T, BNRELU, STORE>(cout, a, s);
  if (loader == BNBWD && epi == MASK) return launch_row_gemm_t<T, BNBWD, MASK>(cout, a, s);
  if (loader == BNBWD && epi == SCATTER) return launch_row_gemm_t<T, BNBWD, SCATTER>(cout, a, s);
  return VLP3D_EINVAL;
}

template <typename T, int LOADER, int COUT, int DYL = BNBWD, bool BFM = false, bool POOL = false>
int launch_wgrad_c(const WgradArgs &w, hipStream_t s, dim3 grid, size_t lds) {
  const int per_wave = ((COUT / 32) * (w.KP / 32) + 3) / 4;
  const dim3 block(256);
  // fewer accumulator registers -> more resident workgroups -> more loads in flight (the kernel is latency bound)
  if (per_wave <= 1) hipLaunchKernelGGL((wgrad_kernel<T, COUT, LOADER, 1, DYL, BFM, POOL>), grid, block, lds, s, w);
  else if (per_wave <= 3) hipLaunchKernelGGL((wgrad_kernel<T, COUT, LOADER, 3, DYL, BFM, POOL>), grid, block, lds, s, w);
  else if (per_wave <= 4) hipLaunchKernelGGL((wgrad_kernel<T, COUT, LOADER, 4, DYL, BFM, POOL>), grid, block, lds, s, w);
  else if (per_wave <= 6) hipLaunchKernelGGL((wgrad_kernel<T, COUT, LOADER, 6, DYL, BFM, POOL>), grid, block, lds, s, w);
  else if (per_wave <= 9) hipLaunchKernelGGL((wgrad_kernel<T, COUT, LOADER, 9, DYL, BFM, POOL>), grid, block, lds, s, w);
  else return VLP3D_EINVAL;
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

template <typename T, int LOADER>
int launch_wgrad_t(int cout, const WgradArgs &w, hipStream_t s) {
  const long long ntiles = w.src.R / 32;
  const long long nblk = (ntiles + w.tiles_per_block - 1) / w.tiles_per_block;
  const dim3 grid((unsigned)nblk);
  const size_t lds = sizeof(T) == 2 ? (size_t)32 * (cout + w.KP + 8) * 2 : (size_t)32 * (cout + w.KP) * sizeof(float);
  if (lds > 64 * 1024) return VLP3D_EINVAL;
  if constexpr (sizeof(T) == 2 && LOADER == BNRELU) {
    if (w.dy.pool_g != nullptr) {  // last layer, bf16 storage: the pooled-gradient loader is its own instantiation
      if (cout == 256) return launch_wgrad_c<T, LOADER, 128, BNBWD, false, true>(w, s, dim3(grid.x, 2), (size_t)32 * (128 + w.KP + 8) * 2);
      if (cout == 128) return launch_wgrad_c<T, LOADER, 128, BNBWD, false, true>(w, s, grid, lds);
      if (cout == 64) return launch_wgrad_c<T, LOADER, 64, BNBWD, false, true>(w, s, grid, lds);
      return VLP3D_EINVAL;
    }
  } else if (sizeof(T) == 2 && w.dy.pool_g != nullptr) {
    return VLP3D_EINVAL;  // the gather layer is never the pooled one
  }
  if (sizeof(T) == 2 && cout == 256) {  // two 128-column halves (see the kernel: coff)
    const size_t lds_half = (size_t)32 * (128 + w.KP + 8) * 2;
    return launch_wgrad_c<T, LOADER, 128>(w, s, dim3(grid.x, 2), lds_half);
  }
  switch (cout) {
    case 64: return launch_wgrad_c<T, LOADER, 64>(w, s, grid, lds);
    case 128: return launch_wgrad_c<T, LOADER, 128>(w, s, grid, lds);
    case 256: return launch_wgrad_c<T, LOADER, 256>(w, s, grid, lds);
    default: return VLP3D_EINVAL;
  }
}

}  // namespace

// ---- C ABI -----------------------------------------------------------------------------------------
// `bf16` selects the storage/MFMA type of Y / G / W (0: fp32, 1: bf16).  Per-channel vectors are fp32.

extern "C" int vlp3d_sa_fwd_gather(const float *xyz, const float *new_xyz, const int *idx, const float *feat_pm,
                                   int B, int N, int M, int S, int C, float radius, const void *W, int K, int cout,
                                   void *Y, double *stats, int bf16_io, const void *crow, const int *rowptr, int nballs, void *stream) {
  if (!xyz || !new_xyz || !idx || !feat_pm || !W || !Y || !stats || B < 1 || N < 1 || M < 1 || S < 1 || C < 4 ||
      (C & 3) || K < C + 4 || (K % (bf16_io ? 16 : 8)) || (((long long)B * M * S) & 31))
    return VLP3D_EINVAL;
  RowGemmArgs a = {};
  a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx; a.feat_pm = feat_pm;
  a.N = N; a.M = M; a.S = S; a.C = C; a.radius = radius;
  if (bf16_io & 2) {  // feat_pm holds bf16 rows of (C + 7) & ~7 columns (zero padded): the gather fast path of the LDS kernels only
    const int kc8 = K / 8;   // the fast path (K <= 160) or the WIDE instantiation (cout 128, K <= 288, C % 8 == 0)
    const bool wide_ok = cout == 128 && 32 * kc8 <= 64 * 18 && kc8 <= 36 && C % 8 == 0;
    if (!(bf16_io & 1) || (K % 16) || !(32 * kc8 <= 64 * 10 || wide_ok) || !(crow || (((long long)M * S) & 31) == 0)) return VLP3D_EINVAL;
    a.feat_bf = feat_pm;
    a.ldf = (C + 7) & ~7;
    bf16_io = 1;
  }
  a.W = W; a.K = K; a.R = (long long)B * M * S; a.Yout = Y; a.ldout = cout; a.stats = stats;
  a.S_shift = (S & (S - 1)) ? -1 : __builtin_ctz(S);
  if (crow) {
    if (!rowptr || nballs < 1 || !bf16_io) return VLP3D_EINVAL;  // compact row map: bf16 LDS kernels only
    a.crow = (const int4 *)crow; a.rowptr = rowptr; a.nballs = nballs;
  }
  a.tile_scene = (((long long)M * S) & 31) == 0;
  return bf16_io ? launch_row_gemm<bf16>(GATHER, STORE, cout, a, (hipStream_t)stream)
                 : launch_row_gemm<float>(GATHER, STORE, cout, a, (hipStream_t)stream);
}

extern "C" int vlp3d_sa_fwd_layer(const void *Yin, long long R, int K, const float *scale, const float *shift,
                                  const void *W, int cout, void *Y, double *stats, int bf16_io, const void *crow, const int *rowptr, int nballs, void *stream) {
  if (!Yin || !scale || !shift || !W || !Y || !stats || R < 32 || (R & 31) || (K % (bf16_io ? 16 : 8)))
    return VLP3D_EINVAL;
  RowGemmArgs a = {};
  a.Yin = Yin; a.ldin = K; a.scale = scale; a.shift = shift;
  a.W = W; a.K = K; a.R = R; a.Yout = Y; a.ldout = cout; a.stats = stats;
  if (crow) {
    if (!rowptr || nballs < 1 || !bf16_io) return VLP3D_EINVAL;  // compact row map: bf16 LDS kernels only
    a.crow = (const int4 *)crow; a.rowptr = rowptr; a.nballs = nballs;
  }
  return bf16_io ? launch_row_gemm<bf16>(BNRELU, STORE, cout, a, (hipStream_t)stream)
                 : launch_row_gemm<float>(BNRELU, STORE, cout, a, (hipStream_t)stream);
}

static int sa_pool_impl(const void *Y, long long BM, int S, int C, const float *scale, const float *shift, float *out,
                        void *out_bf16, unsigned char *sel_idx, int bf16_io, const int *rowptr, void *stream);
extern "C" int vlp3d_sa_pool(const void *Y, long long BM, int S, int C, const float *scale, const float *shift,
                             float *out, unsigned char *sel_idx, int bf16_io, const int *rowptr, void *stream) {
  return sa_pool_impl(Y, BM, S, C, scale, shift, out, nullptr, sel_idx, bf16_io, rowptr, stream);
}
// vlp3d_sa_pool that also writes the pooled rows as bf16 (out_bf16 (BM x C), C % 8 == 0): the next level's gather layer reads them
// through bf16_io bit 1 (include/vlp3d.h)
extern "C" int vlp3d_sa_pool_rows(const void *Y, long long BM, int S, int C, const float *scale, const float *shift,
                                  float *out, void *out_bf16, unsigned char *sel_idx, int bf16_io, const int *rowptr,
                                  void *stream) {
  if (!out_bf16 || (C % 8)) return VLP3D_EINVAL;
  return sa_pool_impl(Y, BM, S, C, scale, shift, out, out_bf16, sel_idx, bf16_io, rowptr, stream);
}
static int sa_pool_impl(const void *Y, long long BM, int S, int C, const float *scale, const float *shift, float *out,
                        void *out_bf16, unsigned char *sel_idx, int bf16_io, const int *rowptr, void *stream) {
  if (!Y || !scale || !shift || !out || !sel_idx || BM < 1 || S < 1 || S > 255 || C < 1) return VLP3D_EINVAL;
  if (rowptr && (C % 8)) return VLP3D_EINVAL;  // compact row map: the 8-channel kernel only
  if (C % 8 == 0) {
    const dim3 grid8((unsigned)((BM * (C / 8) + 255) / 256));
    static const bool wave_form = !(getenv("VLP3D_SA_POOL_WAVE") && atoi(getenv("VLP3D_SA_POOL_WAVE")) == 0);
    if (bf16_io && wave_form && (C == 128 || C == 256) && BM < (1ll << 31)) {
      const dim3 gridw((unsigned)((BM + 3) / 4));
      if (C == 128)
        hipLaunchKernelGGL((pool8_wave_kernel<16>), gridw, dim3(256), 0, (hipStream_t)stream, (const bf16 *)Y, S, BM, scale, shift,
                           out, sel_idx, rowptr, (bf16 *)out_bf16);
      else
        hipLaunchKernelGGL((pool8_wave_kernel<32>), gridw, dim3(256), 0, (hipStream_t)stream, (const bf16 *)Y, S, BM, scale, shift,
                           out, sel_idx, rowptr, (bf16 *)out_bf16);
      VLP3D_LAUNCH_CHECK();
      return VLP3D_OK;
    }
    if (bf16_io)
      hipLaunchKernelGGL((pool8_kernel<bf16>), grid8, dim3(256), 0, (hipStream_t)stream, (const bf16 *)Y, S, C, BM, scale,
                         shift, out, sel_idx, rowptr, (bf16 *)out_bf16);
    else
      hipLaunchKernelGGL((pool8_kernel<float>), grid8, dim3(256), 0, (hipStream_t)stream, (const float *)Y, S, C, BM,
                         scale, shift, out, sel_idx, rowptr, (bf16 *)out_bf16);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  const long long total = BM * C;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (bf16_io)
    hipLaunchKernelGGL((pool_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16 *)Y, C, S, C, BM, scale,
                       shift, out, sel_idx);
  else
    hipLaunchKernelGGL((pool_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float *)Y, C, S, C, BM, scale,
                       shift, out, sel_idx);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_sa_pool_grad(const float *dP, const float *out, const unsigned char *sel_idx, long long BM, int S,
                                  int C, void *G, int bf16_io, void *stream) {
  if (!dP || !out || !sel_idx || !G || BM < 1 || S < 1 || C < 1) return VLP3D_EINVAL;
  const long long total = BM * S * C;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (bf16_io)
    hipLaunchKernelGGL((pool_grad_kernel<bf16>), grid, block, 0, (hipStream_t)stream, dP, out, sel_idx, S, C, BM,
                       (bf16 *)G);
  else
    hipLaunchKernelGGL((pool_grad_kernel<float>), grid, block, 0, (hipStream_t)stream, dP, out, sel_idx, S, C, BM,
                       (float *)G);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// dA_{l-1} = dY_l * W_l   (WT = W_l^T, (K_prev x COUT_l) row-major), dY_l = BN-backward(G_l, Y_l).
// bn5 = [rstd | nmean_rstd | k1 | k2 | k3] of layer l (5 x ld); prev4 = [scale | shift | rstd | nmean_rstd] of
// layer l-1 (4 x kprev).  Writes G_{l-1} (R x kprev) and accumulates tstats (2 x kprev).
extern "C" int vlp3d_sa_bwd_layer(const void *G, const void *Y, long long R, int ld, const float *bn5, const void *WT,
                                  int kprev, const void *Yprev, const float *prev4, void *Gprev, double *tstats,
                                  const float *pool_g, const unsigned char *pool_sel, int pool_S, int bf16_io,
                                  const void *crow, const int *rowptr, int nballs, void *stream) {
  if ((!G && !(pool_g && pool_sel && pool_S > 0)) || !Y || !bn5 || !WT || !Yprev || !prev4 || !Gprev || !tstats || R < 32 || (R & 31) ||
      (ld % (bf16_io ? 16 : 8)))
    return VLP3D_EINVAL;
  RowGemmArgs a = {};
  a.Gin = G; a.Yin = Y; a.ldin = ld;
  a.rstd = bn5; a.nmean_rstd = bn5 + ld; a.k1 = bn5 + 2 * ld; a.k2 = bn5 + 3 * ld; a.k3 = bn5 + 4 * ld;
  a.W = WT; a.K = ld; a.R = R;
  a.Yout = Gprev; a.ldout = kprev; a.Yprev = Yprev; a.ldprev = kprev;
  a.p_scale = prev4; a.p_shift = prev4 + kprev; a.p_rstd = prev4 + 2 * kprev; a.p_nmean_rstd = prev4 + 3 * kprev;
  a.tstats = tstats;
  if (!G) set_pool(a, pool_g, pool_sel, pool_S);
  if (crow) {
    if (!rowptr || nballs < 1 || !bf16_io) return VLP3D_EINVAL;  // compact row map: bf16 LDS kernels only
    a.crow = (const int4 *)crow; a.rowptr = rowptr; a.nballs = nballs;
  }
  return bf16_io ? launch_row_gemm<bf16>(BNBWD, MASK, kprev, a, (hipStream_t)stream)
                 : launch_row_gemm<float>(BNBWD, MASK, kprev, a, (hipStream_t)stream);
}

// first layer: dX = dY_1 * W_1 scattered to d(features) (B,N,C), d(xyz) (B,N,3), d(new_xyz) (B,M,3)
// (each optional; NOT zeroed here).  WT = W_1^T padded to (kpad x COUT_1) with kpad a multiple of 32.
extern "C" int vlp3d_sa_bwd_gather(const void *G, const void *Y, int ld, const float *bn5, const void *WT, int kpad,
                                   const int *idx, int B, int N, int M, int S, int C, float radius, float *dfeat_pm,
                                   float *dxyz, float *dnew_xyz, int bf16_io, const void *crow, const int *rowptr, int nballs, void *stream) {
  if (!G || !Y || !bn5 || !WT || !idx || (kpad & 31) || kpad < C + 3 || (ld % (bf16_io ? 16 : 8)) ||
      (((long long)B * M * S) & 31))
    return VLP3D_EINVAL;
  RowGemmArgs a = {};
  a.Gin = G; a.Yin = Y; a.ldin = ld;
  a.rstd = bn5; a.nmean_rstd = bn5 + ld; a.k1 = bn5 + 2 * ld; a.k2 = bn5 + 3 * ld; a.k3 = bn5 + 4 * ld;
  a.W = WT; a.K = ld; a.R = (long long)B * M * S;
  a.idx = idx; a.N = N; a.M = M; a.S = S; a.C = C; a.radius = radius;
  a.S_shift = (S & (S - 1)) ? -1 : __builtin_ctz(S);
  a.tile_scene = (((long long)M * S) & 31) == 0;
  a.dfeat_pm = dfeat_pm; a.dxyz = dxyz; a.dnew_xyz = dnew_xyz;
  if (crow) {
    if (!rowptr || nballs < 1 || !bf16_io) return VLP3D_EINVAL;  // compact row map: bf16 LDS kernels only
    a.crow = (const int4 *)crow; a.rowptr = rowptr; a.nballs = nballs;
  }
  return bf16_io ? launch_row_gemm<bf16>(BNBWD, SCATTER, kpad, a, (hipStream_t)stream)
                 : launch_row_gemm<float>(BNBWD, SCATTER, kpad, a, (hipStream_t)stream);
}

// dW_l (cout x K) += sum_r dY_l[r]^T A_{l-1}[r];  A_{l-1} = relu(Yprev*scale+shift) (gather == 0) or the gathered
// rows (gather != 0: xyz/new_xyz/idx/feat_pm given, Yprev ignored).  dW must be zeroed by the caller.
extern "C" int vlp3d_sa_wgrad(const void *G, const void *Y, long long R, int cout, const float *bn5, int gather,
                              const void *Yprev, int K, const float *scale, const float *shift, const float *xyz,
                              const float *new_xyz, const int *idx, const float *feat_pm, int N, int M, int S, int C,
                              float radius, float *dW, float *partials, int max_blocks, const float *pool_g,
                              const unsigned char *pool_sel, int pool_S, int bf16_io, int defer_reduce, const void *crow,
                              const int *rowptr, int nballs, void *stream) {
  if ((!G && !(pool_g && pool_sel && pool_S > 0)) || !Y || !bn5 || (!dW && !defer_reduce) || !partials || max_blocks < 1 || R < 32 || (R & 31) || R >= (1ll << 31) || K < 1 || (K & 3)) return VLP3D_EINVAL;
  WgradArgs w = {};
  w.src.K = K; w.src.R = R;
  if (gather) {
    if (!xyz || !new_xyz || !idx || !feat_pm || M < 1 || S < 1 || (((long long)M * S) & 31)) return VLP3D_EINVAL;
    w.src.xyz = xyz; w.src.new_xyz = new_xyz; w.src.idx = idx; w.src.feat_pm = feat_pm;
    w.src.N = N; w.src.M = M; w.src.S = S; w.src.C = C; w.src.radius = radius;
    if (bf16_io & 2) {  // bf16 feature rows, as in vlp3d_sa_fwd_gather
      if (!(bf16_io & 1)) return VLP3D_EINVAL;
      w.src.feat_bf = feat_pm;
      w.src.ldf = (C + 7) & ~7;
    }
    w.src.S_shift = (S & (S - 1)) ? -1 : __builtin_ctz(S);
  } else {
    if (!Yprev || !scale || !shift || (256 % (K / 4)) || ((K / 4) & (K / 4 - 1))) return VLP3D_EINVAL;  // fixed staging columns per thread, shift-indexed rows
    w.src.Yin = Yprev; w.src.ldin = K; w.src.scale = scale; w.src.shift = shift;
  }
  w.dy.Gin = G; w.dy.Yin = Y; w.dy.ldin = cout;
  if (!G) set_pool(w.dy, pool_g, pool_sel, pool_S);
  w.dy.rstd = bn5; w.dy.nmean_rstd = bn5 + cout; w.dy.k1 = bn5 + 2 * cout; w.dy.k2 = bn5 + 3 * cout;
  w.dy.k3 = bn5 + 4 * cout;
  if (crow) {  // compact row map: both loaders (the gathered / previous-layer operand and dY) index compact rows
    if (!rowptr || nballs < 1 || !bf16_io) return VLP3D_EINVAL;
    w.src.crow = w.dy.crow = (const int4 *)crow;
    w.src.rowptr = w.dy.rowptr = rowptr;
    w.src.nballs = w.dy.nballs = nballs;
  }
  w.KP = (K + 31) & ~31;
  w.partials = partials;
  const long long ntiles = R / 32;
  long long tpb = (ntiles + max_blocks - 1) / max_blocks;  // each workgroup accumulates its rows in registers
  if (tpb < 1) tpb = 1;
  w.tiles_per_block = tpb;
  const int nblk = (int)((ntiles + tpb - 1) / tpb);
  hipStream_t s = (hipStream_t)stream;
  int st;
  if (bf16_io) st = gather ? launch_wgrad_t<bf16, GATHER>(cout, w, s) : launch_wgrad_t<bf16, BNRELU>(cout, w, s);
  else st = gather ? launch_wgrad_t<float, GATHER>(cout, w, s) : launch_wgrad_t<float, BNRELU>(cout, w, s);
  if (st != VLP3D_OK) return st;
  if (defer_reduce) return VLP3D_OK;  // the slabs are summed later by vlp3d_slab_reduce_batch
  const int n = cout * K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, s, partials, nblk, n, dW);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}


// vec (4 x C) f32 = [scale | shift | rstd | -mean*rstd] from the fp64 batch sums (training) or the running
// statistics (eval); in training also updates running_mean / running_var (may be NULL) with `momentum`.
extern "C" int vlp3d_sa_stat_slabs(long long R) { return (int)grid_tiles(R); }

extern "C" int vlp3d_sa_bn_fold(const double *stats, int nslab, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, int C, long long R, float eps, float momentum,
                                int training, float *vec, void *stream) {
  if (nslab < 1 || !gamma || !beta || !vec || C < 1 || R < 1 || (training && !stats) || (!training && (!running_mean || !running_var)))
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + SUMC - 1) / SUMC), dim3(256), 0, (hipStream_t)stream, stats, nslab, gamma, beta,
                     running_mean, running_var, C, (double)R, eps, momentum, training, vec, (const float *)nullptr);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// as vlp3d_sa_bn_fold, the running mean additionally tracking `mean_shift` (C): the bias of a Conv1d in front of a
// train-mode BatchNorm1d (voting_module.py:41-44), which the rows kernels leave out of Y because it cancels in the output.
extern "C" int vlp3d_sa_bn_fold_shift(const double *stats, int nslab, const float *gamma, const float *beta,
                                      float *running_mean, float *running_var, int C, long long R, float eps, float momentum,
                                      int training, float *vec, const float *mean_shift, void *stream) {
  if (nslab < 1 || !gamma || !beta || !vec || C < 1 || R < 1 || (training && !stats) || (!training && (!running_mean || !running_var)))
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + SUMC - 1) / SUMC), dim3(256), 0, (hipStream_t)stream, stats, nslab, gamma, beta,
                     running_mean, running_var, C, (double)R, eps, momentum, training, vec, mean_shift);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// bn5 (5 x C) backward constants + dgamma, dbeta (C) from vec, gamma and the reductions t (2 x C) f64.
extern "C" int vlp3d_sa_bn_bwd_consts(const float *vec, const float *gamma, const double *t, int nslab, int C,
                                      long long R, int training, float *bn5, float *dgamma, float *dbeta,
                                      void *stream) {
  if (nslab < 1 || !vec || !gamma || !t || !bn5 || !dgamma || !dbeta || C < 1 || R < 1) return VLP3D_EINVAL;
  hipLaunchKernelGGL(bn5_kernel, dim3((C + SUMC - 1) / SUMC), dim3(256), 0, (hipStream_t)stream, vec, gamma, t, nslab, C, (double)R,
                     training, bn5, dgamma, dbeta);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// t (slabs x 2 x C) f64 = per-workgroup [sum g, sum g*yhat] of the last layer from the pooled tensors: one slab per row
// group (the fp64 atomics into one (2 x C) buffer needed a clear launch in front of every SA backward).
static long long pool_tstats_rows_per_block(long long BM) {
  const long long rpb = (BM + 127) / 128;
  return rpb < 16 ? 16 : rpb;
}
extern "C" int vlp3d_sa_pool_tstats_slabs(long long BM) {
  if (BM < 1) return 0;
  const long long rpb = pool_tstats_rows_per_block(BM);
  return (int)((BM + rpb - 1) / rpb);
}
extern "C" int vlp3d_sa_pool_tstats(const float *dP, const float *out, const float *gamma, const float *beta,
                                    long long BM, int C, double *t, float *gsel, void *stream) {
  if (!dP || !out || !gamma || !beta || !t || !gsel || BM < 1 || C < 1) return VLP3D_EINVAL;
  const long long rpb = pool_tstats_rows_per_block(BM);
  const dim3 grid((C + 255) / 256, (unsigned)((BM + rpb - 1) / rpb));
  hipLaunchKernelGGL(pool_tstats_kernel, grid, dim3(256), 0, (hipStream_t)stream, dP, out, gamma, beta, BM, C, rpb, t,
                     gsel);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}


// ---- plain linear layers on the same kernels (fp32 storage, exact-fp32 MFMA) -------------------------------------
// The transformer / head projections of this path are 2048..16384-row x 128..256 GEMMs; the BLAS library runs
// them at ~5 TFLOP/s (99 us for 16384x128x128).  row_gemm<PLAIN, BIAS> does the same product in ~15 us.

// Y (R x N) = X (R x K) W^T + bias;  W (N x K) row-major, bias (N) or NULL.  R % 32 == 0, K % 8 == 0,
// N in {32, 64, 128, 160, 256, 288}.
// csrc/linear_tile.hip: LDS-tiled form for many rows (bf16 operands)
int vlp3d_internal_linear_tile(const float *X, int ldx, const float *W, int ldw, int kdim, int ncols, const float *bias,
                               const float *base, long long R, float *Y, int ldy, int transposed_weight, hipStream_t stream);
static const long long LINEAR_TILE_MIN_ROWS = 8192;  // below: a wave per 32 x 32 tile fills the chip better (latency bound)

extern "C" int vlp3d_linear_fwd(const float *X, const float *W, const float *bias, long long R, int K, int N, float *Y,
                                int bf16_mma, void *stream) {
  if (!X || !W || !Y || R < 32 || (R & 31) || K < 8 || (K & 7)) return VLP3D_EINVAL;
  if (bf16_mma && R >= LINEAR_TILE_MIN_ROWS && N % 64 == 0 && K % 16 == 0)
    return vlp3d_internal_linear_tile(X, K, W, K, K, N, bias, nullptr, R, Y, N, 0, (hipStream_t)stream);
  if (bf16_mma && R <= 65536 && N % 32 == 0 && N >= 32 && K % 16 == 0) {
    hipLaunchKernelGGL((linear_bf16_kernel<false>), dim3(grid_tiles(R), N / 32), dim3(256), 0, (hipStream_t)stream, X, K, W, K, 0,
                       bias, nullptr, R, Y, N);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  RowGemmArgs a = {};
  a.Yin = X; a.ldin = K; a.W = W; a.K = K; a.R = R; a.Yout = Y; a.ldout = N; a.scale = bias;
  if (R <= 65536 && N % 32 == 0 && N >= 64 && N <= 1024) {
    // few rows: a wave per 32 x 32 output tile (4x..8x more waves than a wave per 32 x N) — these calls are
    // latency bound, the re-read of the A rows comes from L2
    hipLaunchKernelGGL((row_gemm_kernel<float, 32, PLAIN, BIAS>), dim3(grid_tiles(R), N / 32), dim3(256), 0,
                       (hipStream_t)stream, a);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  return launch_row_gemm_t<float, PLAIN, BIAS>(N, a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// All weight layouts of one fused SA layer stack in ONE launch (was: cat + zero-fill + 3 casts forward, two
// transposes + zero-fill + slice copy backward — nine tiny launches per SA module and step):
//   W1p (c0 x K1)  = [W1[:, 3:3+C] | W1[:, 0:3] | 0]      (first-layer weight in the row layout [features | xyz | 0])
//   W2d (c1 x c0), W3d (c2 x c1)                           (cast only)
//   WT1 (kpad x c0) = W1p^T zero-padded to kpad rows,  WT2 (c0 x c1) = W2^T,  WT3 (c1 x c2) = W3^T   (backward)
// `out` is one buffer holding the six matrices back to back in that order.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sa_prep_weights_kernel(const float *__restrict__ W1, const float *__restrict__ W2,
                                                              const float *__restrict__ W3, int C, int c0, int c1, int c2,
                                                              int K1, int kpad, T *__restrict__ out) {
  const int n1 = c0 * K1, n2 = c1 * c0, n3 = c2 * c1, n4 = kpad * c0, n5 = c0 * c1, n6 = c1 * c2;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n1 + n2 + n3 + n4 + n5 + n6) return;
  const int o = i;
  float v;
  auto w1p = [&](int row, int k) -> float {  // W1p[row][k]
    if (k < C) return W1[row * (C + 3) + 3 + k];
    if (k < C + 3) return W1[row * (C + 3) + (k - C)];
    return 0.f;
  };
  if (i < n1) {
    v = w1p(i / K1, i % K1);
  } else if ((i -= n1) < n2) {
    v = W2[i];
  } else if ((i -= n2) < n3) {
    v = W3[i];
  } else if ((i -= n3) < n4) {
    const int k = i / c0, row = i - k * c0;  // WT1[k][row] = W1p[row][k]
    v = k < K1 ? w1p(row, k) : 0.f;
  } else if ((i -= n4) < n5) {
    const int k = i / c1, row = i - k * c1;  // WT2[k][row] = W2[row][k]
    v = W2[row * c0 + k];
  } else {
    i -= n5;
    const int k = i / c2, row = i - k * c2;  // WT3[k][row] = W3[row][k]
    v = W3[row * c1 + k];
  }
  st1(out + o, v);
}

extern "C" int vlp3d_sa_prep_weights(const float *W1, const float *W2, const float *W3, int C, int c0, int c1, int c2,
                                     int K1, int kpad, void *out, int bf16_io, void *stream) {
  if (!W1 || !W2 || !W3 || !out || C < 1 || c0 < 1 || c1 < 1 || c2 < 1 || K1 < C + 3 || kpad < K1) return VLP3D_EINVAL;
  const long long total = (long long)c0 * K1 + (long long)c1 * c0 + (long long)c2 * c1 + (long long)kpad * c0 +
                          (long long)c0 * c1 + (long long)c1 * c2;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (bf16_io)
    hipLaunchKernelGGL(sa_prep_weights_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, W1, W2, W3, C, c0, c1, c2,
                       K1, kpad, (bf16 *)out);
  else
    hipLaunchKernelGGL(sa_prep_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, W1, W2, W3, C, c0, c1, c2,
                       K1, kpad, (float *)out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// dX (R x K) = dY (R x N) * W, W (N x K) row-major as stored by nn.Linear — no transposed copy of the weight.
extern "C" int vlp3d_linear_dgrad(const float *dY, const float *W, long long R, int N, int K, float *dX, const float *base,
                                  int bf16_mma, void *stream) {
  if (!dY || !W || !dX || R < 32 || (R & 31) || N < 8 || (N & 7) || K < 32 || (K & 31)) return VLP3D_EINVAL;
  if (base && !(bf16_mma && N % 16 == 0)) return VLP3D_EINVAL;  // the exact-fp32 form has no fused add: add it yourself
  if (bf16_mma && N % 16 == 0 && R >= LINEAR_TILE_MIN_ROWS && K % 64 == 0)
    return vlp3d_internal_linear_tile(dY, N, W, K, N, K, nullptr, base, R, dX, K, 1, (hipStream_t)stream);
  if (bf16_mma && N % 16 == 0) {
    hipLaunchKernelGGL((linear_bf16_kernel<true>), dim3(grid_tiles(R), K / 32), dim3(256), 0, (hipStream_t)stream, dY, N, W, N, K,
                       nullptr, base, R, dX, K);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  RowGemmArgs a = {};
  a.Yin = dY; a.ldin = N; a.W = W; a.K = N; a.ldw = K; a.R = R; a.Yout = dX; a.ldout = K;
  hipLaunchKernelGGL((row_gemm_kernel<float, 32, PLAIN, BIAS_WT>), dim3(grid_tiles(R), K / 32), dim3(256), 0,
                     (hipStream_t)stream, a);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// dW (N x K) = dY^T X   (dY (R x N), X (R x K));  partials: max_blocks * N * K floats of scratch.
// with_bias != 0: dW has N*K + N elements, the last N are the bias gradient sum_r dY[r][:]; partials then holds
// max_blocks * (N*K + N) floats.
extern "C" int vlp3d_linear_wgrad(const float *dY, const float *X, long long R, int K, int N, float *dW, float *partials,
                                  int max_blocks, int with_bias, int defer_reduce, int bf16_mma, void *stream) {
  if (!dY || !X || (!dW && !defer_reduce) || !partials || max_blocks < 1 || R < 32 || (R & 31) || R >= (1ll << 31) || K < 4 || (K & 3) ||
      ((K / 4) & (K / 4 - 1)) || (N & 31))  // K/4 a power of two: the staging row index is a shift
    return VLP3D_EINVAL;
  WgradArgs w = {};
  w.colsum = with_bias != 0;
  w.src.K = K; w.src.R = R; w.src.Yin = X; w.src.ldin = K;
  w.dy.Yin = dY; w.dy.ldin = N;
  w.KP = (K + 31) & ~31;
  w.partials = partials;
  const long long ntiles = R / 32;
  long long tpb = (ntiles + max_blocks - 1) / max_blocks;
  if (tpb < 1) tpb = 1;
  w.tiles_per_block = tpb;
  const int nblk = (int)((ntiles + tpb - 1) / tpb);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nblk);
  int st;
  const size_t esz = bf16_mma ? 2 : 4;  // staged element size
  const int pad = bf16_mma ? 8 : 0;     // the bf16 tiles carry 4 shorts of row padding each
  if (bf16_mma && N % 64 == 0 && N <= 512 && w.KP <= 256) {
    // 64-column workgroup blocks: two to eight times the workgroups for the same slab volume (these launches have few
    // row tiles; with one workgroup per 128 rows and all columns half of the CUs stayed idle)
    const size_t lds64 = (size_t)32 * (64 + w.KP + pad) * esz;
    st = launch_wgrad_c<float, PLAIN, 64, PLAIN, true>(w, s, dim3((unsigned)nblk, N / 64), lds64);
  } else if ((N > 256 || (N / 32) * (w.KP / 32) > 36) && N % 128 == 0 && N <= 1024) {
    // wide layers (merged q/k/v) or more than 9 output tiles per wave: 128-column workgroup blocks
    const size_t lds128 = (size_t)32 * (128 + w.KP + pad) * esz;
    if (lds128 > 64 * 1024) return VLP3D_EINVAL;
    const dim3 g2((unsigned)nblk, N / 128);
    st = bf16_mma ? launch_wgrad_c<float, PLAIN, 128, PLAIN, true>(w, s, g2, lds128)
                  : launch_wgrad_c<float, PLAIN, 128, PLAIN>(w, s, g2, lds128);
  } else {
    const size_t lds = (size_t)32 * (N + w.KP + pad) * esz;
    if (lds > 64 * 1024) return VLP3D_EINVAL;
    switch (N) {
      case 64: st = bf16_mma ? launch_wgrad_c<float, PLAIN, 64, PLAIN, true>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 64, PLAIN>(w, s, grid, lds); break;
      case 128: st = bf16_mma ? launch_wgrad_c<float, PLAIN, 128, PLAIN, true>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 128, PLAIN>(w, s, grid, lds); break;
      case 256: st = bf16_mma ? launch_wgrad_c<float, PLAIN, 256, PLAIN, true>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 256, PLAIN>(w, s, grid, lds); break;
      default: return VLP3D_EINVAL;
    }
  }
  if (st != VLP3D_OK) return st;
  if (defer_reduce) return VLP3D_OK;
  const int n = N * K + (with_bias ? N : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, s, partials, nblk, n, dW);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

namespace {
template <int LOADER, int DYL, bool XB16 = false>
int launch_rows_batch(int maxt, const RowsWgradBatch &t, dim3 grid, size_t lds, hipStream_t s) {
  switch (maxt) {
    case 1: hipLaunchKernelGGL((rows_wgrad_batch_kernel<LOADER, DYL, 1, XB16>), grid, dim3(256), lds, s, t); break;
    case 3: hipLaunchKernelGGL((rows_wgrad_batch_kernel<LOADER, DYL, 3, XB16>), grid, dim3(256), lds, s, t); break;
    case 4: hipLaunchKernelGGL((rows_wgrad_batch_kernel<LOADER, DYL, 4, XB16>), grid, dim3(256), lds, s, t); break;
    case 6: hipLaunchKernelGGL((rows_wgrad_batch_kernel<LOADER, DYL, 6, XB16>), grid, dim3(256), lds, s, t); break;
    default: return VLP3D_EINVAL;
  }
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
}  // namespace

// vlp3d_rows_wgrad (bf16_mma = 1, defer_reduce = 1) for `count` layers at once: jobs of one kernel instantiation (operand
// loader, dY loader, accumulator tiles per wave) share a launch; every job writes its own slabs (jobs[i].partials, same
// layout and slab count as the single entry gives for jobs[i].max_blocks); the caller sums them with
// vlp3d_slab_reduce_batch afterwards.  N % 64 == 0, N <= 512, K as for vlp3d_rows_wgrad (a plain linear layer: K % 4 == 0,
// K / 4 a power of two, K <= 256).
extern "C" int vlp3d_rows_wgrad_batch(const vlp3d_rows_wgrad_job *jobs, int count, void *stream) {
  if (!jobs || count < 1 || count > 1024) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static thread_local int key[1024];
  static thread_local bool done[1024];
  for (int i = 0; i < count; ++i) {
    const vlp3d_rows_wgrad_job &q = jobs[i];
    const int K = q.K, N = q.N;
    const long long R = q.R;
    if (!q.G || !q.X || !q.partials || q.max_blocks < 1 || R < 32 || (R & 31) || R >= (1ll << 31) || K < 4 || (K & 3) ||
        ((K / 4) & (K / 4 - 1)) || (256 % (K / 4)) || K > 288 || N < 64 || (N & 63) || N > 512 || q.ldg < N || q.lda < K ||
        (q.bn5 && (!q.Ypre || q.with_bias)) || (q.a_scale && !q.a_shift) || (q.x_bf16 && (q.bn5 || q.a_scale || (q.lda & 3))))
      return VLP3D_EINVAL;
    const int KP = (K + 31) & ~31;
    const int per_wave = (2 * (KP / 32) + 3) / 4;  // output tiles per wave of a 64-column block
    // accumulator buckets 1 / 4 / 6: a job runs in any instantiation with at least its tile count, and every bucket is one more
    // serialised launch of a few hundred workgroups (with a bucket 3 the K = 128 and K = 256 layers shared nothing: 8 launches
    // per step instead of 4, 4.42 -> 4.39 ms; all four loader combinations behind ONE launch — a workgroup-uniform branch
    // over four inlined bodies — was measured too: 4.42 -> 4.43 ms, not kept)
    const int maxt = per_wave <= 1 ? 1 : (per_wave <= 4 ? 4 : 6);
    key[i] = ((q.x_bf16 ? 1 : 0) << 12) | ((q.a_scale ? 1 : 0) << 8) | ((q.bn5 ? 1 : 0) << 4) | maxt;
    done[i] = false;
  }
  for (int i = 0; i < count; ++i) {
    if (done[i]) continue;
    RowsWgradBatch t;
    int n = 0, gx = 0, gy = 0, kp = 0;
    for (int k = i; k < count && n < ROWS_WGRAD_BATCH; ++k) {
      if (done[k] || key[k] != key[i]) continue;
      const vlp3d_rows_wgrad_job &q = jobs[k];
      const long long ntiles = q.R / 32;
      long long tpb = (ntiles + q.max_blocks - 1) / q.max_blocks;
      if (tpb < 1) tpb = 1;
      RowsWgradJob &j = t.j[n];
      j.G = q.G; j.Ypre = q.Ypre; j.bn5 = q.bn5; j.X = q.X; j.a_scale = q.a_scale; j.a_shift = q.a_shift;
      j.partials = q.partials;
      j.R = (int)q.R; j.K = q.K; j.N = q.N; j.ldg = q.ldg; j.lda = q.lda;
      j.nblk = (int)((ntiles + tpb - 1) / tpb);
      j.tpb = (int)tpb;
      j.colsum = q.with_bias != 0;
      const int KP = (q.K + 31) & ~31;
      if (j.nblk > gx) gx = j.nblk;
      if (q.N / 64 > gy) gy = q.N / 64;
      if (KP > kp) kp = KP;
      done[k] = true;
      ++n;
    }
    const int xb = key[i] >> 12, relu = (key[i] >> 8) & 1, bn = (key[i] >> 4) & 1, maxt = key[i] & 15;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)n);
    const size_t lds = (size_t)32 * (64 + kp + 8) * 2;
    int st;
    if (xb) st = launch_rows_batch<PLAIN, PLAIN, true>(maxt, t, grid, lds, s);  // (host-checked: plain linear jobs only)
    else if (bn) st = relu ? launch_rows_batch<BNRELU, BNBWD>(maxt, t, grid, lds, s) : launch_rows_batch<PLAIN, BNBWD>(maxt, t, grid, lds, s);
    else st = relu ? launch_rows_batch<BNRELU, PLAIN>(maxt, t, grid, lds, s) : launch_rows_batch<PLAIN, PLAIN>(maxt, t, grid, lds, s);
    if (st != VLP3D_OK) return st;
  }
  return VLP3D_OK;
}

// The same for plain linear layers given as vlp3d_linear_wgrad's arguments (K <= 256).
extern "C" int vlp3d_linear_wgrad_batch(const vlp3d_linear_wgrad_job *jobs, int count, void *stream) {
  if (!jobs || count < 1 || count > 1024) return VLP3D_EINVAL;
  static thread_local vlp3d_rows_wgrad_job r[1024];
  for (int i = 0; i < count; ++i) {
    const vlp3d_linear_wgrad_job &q = jobs[i];
    if (q.K > 256) return VLP3D_EINVAL;
    r[i] = vlp3d_rows_wgrad_job{q.dY, nullptr, q.N, nullptr, q.X, q.K, nullptr, nullptr, q.R, q.K, q.N, q.partials,
                                q.max_blocks, q.with_bias, q.x_bf16};
  }
  return vlp3d_rows_wgrad_batch(r, count, stream);
}

// Weight gradient of a layer of a rows stack (csrc/rows_mlp.hip): dW[:, 0:K] (N x K, row stride ldo) = dY^T A over R rows,
//   dY = G (R x N, row stride ldg)                      when bn5 == NULL   (a plain linear layer; dbias = column sums of G)
//      = BatchNorm-backward of (G, Ypre) with bn5 [5][N] otherwise       (dbias must be NULL)
//   A  = X (R x K, row stride lda)                      when a_scale == NULL
//      = relu(X * a_scale + a_shift)                    otherwise (the previous layer's folded BatchNorm, length K)
// N % 64 == 0, K % 32 == 0, K <= 288, 256 % (K/4) == 0; wider K: call per K-slice (offset X / a_scale / a_shift / dW).
// partials: max_blocks * (N*K + N) floats of scratch.
extern "C" int vlp3d_rows_wgrad(const float *G, const float *Ypre, int ldg, const float *bn5, const float *X, int lda,
                                const float *a_scale, const float *a_shift, long long R, int K, int N, float *dW, int ldo,
                                float *dbias, float *partials, int max_blocks, int defer_reduce, int bf16_mma, void *stream) {
  if (!G || !X || (!dW && !defer_reduce) || !partials || max_blocks < 1 || R < 32 || (R & 31) || R >= (1ll << 31) || K < 32 || (K & 31) ||
      K > 288 || (256 % (K / 4)) || ((K / 4) & (K / 4 - 1)) || N < 64 || (N & 63) || ldg < N || lda < K || ldo < K ||
      (bn5 && (!Ypre || dbias)) || (a_scale && !a_shift))
    return VLP3D_EINVAL;
  WgradArgs w = {};
  w.colsum = dbias != nullptr;  // deferred: a non-NULL dbias only asks for the column sums in the slabs
  w.src.K = K; w.src.R = R; w.src.Yin = X; w.src.ldin = lda; w.src.scale = a_scale; w.src.shift = a_shift;
  w.dy.ldin = ldg;
  if (bn5) {
    w.dy.Gin = G; w.dy.Yin = Ypre;
    w.dy.rstd = bn5; w.dy.nmean_rstd = bn5 + N; w.dy.k1 = bn5 + 2 * N; w.dy.k2 = bn5 + 3 * N; w.dy.k3 = bn5 + 4 * N;
  } else {
    w.dy.Yin = G;
  }
  w.KP = K;
  w.partials = partials;
  const long long ntiles = R / 32;
  long long tpb = (ntiles + max_blocks - 1) / max_blocks;
  if (tpb < 1) tpb = 1;
  w.tiles_per_block = tpb;
  const int nblk = (int)((ntiles + tpb - 1) / tpb);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nblk, N / 64);
  const size_t lds = bf16_mma ? (size_t)32 * (64 + K + 8) * 2 : (size_t)32 * (64 + K) * sizeof(float);
  int st;
  if (bf16_mma) {
    if (bn5) st = a_scale ? launch_wgrad_c<float, BNRELU, 64, BNBWD, true>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 64, BNBWD, true>(w, s, grid, lds);
    else st = a_scale ? launch_wgrad_c<float, BNRELU, 64, PLAIN, true>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 64, PLAIN, true>(w, s, grid, lds);
  } else {
    if (bn5) st = a_scale ? launch_wgrad_c<float, BNRELU, 64, BNBWD>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 64, BNBWD>(w, s, grid, lds);
    else st = a_scale ? launch_wgrad_c<float, BNRELU, 64, PLAIN>(w, s, grid, lds) : launch_wgrad_c<float, PLAIN, 64, PLAIN>(w, s, grid, lds);
  }
  if (st != VLP3D_OK) return st;
  if (defer_reduce) return VLP3D_OK;
  const int n = N * K + (dbias ? N : 0);
  hipLaunchKernelGGL(wgrad_reduce_strided_kernel, dim3((n + 15) / 16), dim3(256), 0, s, partials, nblk, N, K, dbias ? 1 : 0, dW,
                     ldo, dbias);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// Sum the per-workgroup slabs of `count` deferred weight-gradient launches (descs: HOST array, read during the call).
extern "C" int vlp3d_slab_reduce_batch(const vlp3d_slab_reduce_desc *descs, int count, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int c0 = 0; c0 < count; c0 += SLAB_BATCH) {
    SlabBatch t = {};
    t.count = count - c0 < SLAB_BATCH ? count - c0 : SLAB_BATCH;
    int blocks = 0;
    for (int j = 0; j < t.count; ++j) {
      const vlp3d_slab_reduce_desc &d = descs[c0 + j];
      if (!d.partials || !d.dst || d.nblk < 1 || d.n_mat < 4 || (d.n_mat & 3) || d.n_bias < 0 || (d.n_bias & 3) ||
          (d.n_bias && !d.dbias) || d.K < 4 || (d.K & 3) || (d.n_mat % d.K) || d.ldo < (d.ncol_out > 0 ? d.ncol_out : d.K) ||
          d.ncol_out > d.K || d.rot < 0 || (((size_t)d.partials) & 15))
        return VLP3D_EINVAL;
      t.d[j] = d;
      t.first_block[j] = blocks;
      blocks += (d.n_mat + d.n_bias + 63) / 64;
    }
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t);
    VLP3D_LAUNCH_CHECK();
  }
  return VLP3D_OK;
}
