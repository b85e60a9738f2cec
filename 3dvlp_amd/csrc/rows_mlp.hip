// Exact-fp32 dense stacks on FEW rows (R <= 64K): the 1x1 Conv(+bias) -> BatchNorm -> ReLU chains of the feature-
// propagation modules (lib/pointnet2/pointnet2_modules.py:403-416), the voting module (voting_module.py:33-60) and the ROI
// heads (roi_heads.py:15-147) as matrix-core products on ROW-MAJOR (point-major) matrices — no NCHW convolutions, layout
// transposes, library BatchNorm or separate ReLU kernels.  Same arithmetic as the grouped per-ball MLP (csrc/sa_mlp.hip):
// train-mode BatchNorm = column statistics of the pre-activation Y, folded into per-channel scale/shift that the NEXT
// product applies while loading its A operand; backward folds BatchNorm/ReLU backward into loaders and epilogues.
//
// One wave owns a 32 x 32 output tile (few rows: a wave per tile gives 4-8x more waves than a wave per 32 x N) and walks
// the whole K with v_mfma_f32_32x32x2_f32 (exact fp32: these layers are part of the 1e-4 parity gate).
//   LOADER  PLAIN   a = X[r][k]
//           BNRELU  a = relu(X[r][k] * scale[k] + shift[k])                (X = previous pre-activation)
//           BNBWD   a = k1[k] * (G[r][k] - k2[k] - (X[r][k]*rstd[k] + nmr[k]) * k3[k])   (BatchNorm backward of (G, Y = X))
//   EPI     STORE   Y = acc, per-column sum / sum of squares -> this workgroup's slab (fp64, no atomics)
//           BIAS    Y = acc + bias (bias may be NULL)
//           MASK    G = relu'(Yprev*p_scale + p_shift) * acc, column sums of G and G * yhat_prev -> slab
// The weight is (N x K) row-major, or K-major (wt = 1: (K x ldw), i.e. the SAME buffer read as the transposed operand of
// the input-gradient product — no transposed copies).
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

enum Loader { PLAIN = 0, BNRELU = 1, BNBWD = 2 };
enum Epilogue { STORE = 0, BIAS = 1, MASK = 2 };

struct RowsArgs {
  const float *X, *G;  // A operand source(s), row stride ldx
  int ldx;
  const float *a_scale, *a_shift;  // BNRELU
  const float *bn5;                // BNBWD: [rstd | -mean*rstd | gamma*rstd | mean(g) | mean(g*yhat)], each of length K
  const float *W;
  int ldw, wt;
  int bfm;  // bf16 MFMA operands (rounded in registers), fp32 I/O and accumulation: the timing configuration
  int wlds; // bfm && !wt: the workgroup's 128 x K weight block is staged ONCE in LDS as bf16 (set by launch() when it fits)
  int K, N;
  long long R;
  float *Y;
  int ldy;
  const float *bias;
  double *stats;  // STORE / MASK: slabs [gridDim.x][2][N]
  const float *Yprev;
  int ldprev;
  const float *p_vec;  // MASK: [scale | shift | rstd | -mean*rstd] of the previous BatchNorm, each of length N
};

__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// A workgroup (4 waves) owns a 32-row x 128-column output tile.  The 32 x K A tile is loaded ONCE, cooperatively and
// coalesced (consecutive lanes read consecutive 16-byte chunks of a row; a lane always stages the same four columns, so
// the loader's per-column constants are fetched once per workgroup), transformed, and kept in LDS (rows padded by 16 B:
// conflict-free ds_read_b128 fragment reads); wave w then walks K for columns 32w..32w+31 with its W fragments
// double-buffered in registers, 32 columns of K ahead.  (First version: a wave per 32 x 32 tile with every operand
// straight from memory — one exposed memory latency per 8 columns of K, 45 us for 8192 x 512 x 256; register-level
// double buffering of both operands: 42 us; the products themselves are 7 us per wave.)
struct WChunk {
  float4 w[4];
};

__device__ __forceinline__ void w_load(const RowsArgs &a, int col, int kbase, int half, WChunk &c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k0 = kbase + 8 * s + 4 * half;  // lane half h covers columns 8s + 4h .. +3 of the chunk
    if (a.wt) {  // K-major weight: four dword loads, each coalesced over the lanes' 32 output columns
      const float *wp = a.W + (long long)k0 * a.ldw + col;
      c.w[s] = make_float4(wp[0], wp[a.ldw], wp[2 * a.ldw], wp[3 * a.ldw]);
    } else {
      c.w[s] = ld4(a.W + (long long)col * a.ldw + k0);
    }
  }
}

typedef __attribute__((__vector_size__(8 * sizeof(short)))) short bf16x8_t;
__device__ __forceinline__ short bf16_bits_(float v) {
  __hip_bfloat16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}
__device__ __forceinline__ bf16x8_t pack_bf16x8(const float4 &a, const float4 &b) {
  bf16x8_t v;
  v[0] = bf16_bits_(a.x); v[1] = bf16_bits_(a.y); v[2] = bf16_bits_(a.z); v[3] = bf16_bits_(a.w);
  v[4] = bf16_bits_(b.x); v[5] = bf16_bits_(b.y); v[6] = bf16_bits_(b.z); v[7] = bf16_bits_(b.w);
  return v;
}

__device__ __forceinline__ void w_mma(const float *sa, int kbase, int half, const WChunk &c, f32x16 &acc, int bfm) {
  if (bfm) {  // two 32x32x16 bf16 steps per 32 columns of K: step j contracts the k's of chunks 2j and 2j+1 (any pairing of
              // the summation index works as long as both operands use the same one)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 a0 = *reinterpret_cast<const float4 *>(sa + kbase + 16 * j + 4 * half);
      const float4 a1 = *reinterpret_cast<const float4 *>(sa + kbase + 16 * j + 8 + 4 * half);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_bf16x8(a0, a1), pack_bf16x8(c.w[2 * j], c.w[2 * j + 1]), acc, 0, 0, 0);
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 av = *reinterpret_cast<const float4 *>(sa + kbase + 8 * s + 4 * half);
    const float4 bv = c.w[s];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
}

template <int LOADER, int EPI>
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(RowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sA[];  // [32][K + 4]; then, with a.wlds, bf16 sW[128][K + 4]
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = blockIdx.y * 128 + 32 * wave, col = c0 + r;  // this wave's 32 output columns
  const bool active = c0 < a.N;                               // wave-uniform (N % 32 == 0)
  const int lds_ld = a.K + 4, kq = a.K / 4;                   // kq = 16-byte chunks per A row
  const long long ntiles = a.R / 32;
  // staging: chunk e = threadIdx.x + 256 j of the tile's 32 * kq 16-byte chunks -> (row e / kq, columns 4 (e % kq) ..).
  // BatchNorm loaders need 256 % kq == 0 (host-checked): a thread then always stages the same four columns and its
  // per-column constants are fetched once.
  const int nch = 32 * kq, scol = (threadIdx.x % kq) * 4;
  float4 k_a, k_b, k_c, k_d, k_e;
  if (LOADER == BNRELU) {
    k_a = ld4(a.a_scale + scol);
    k_b = ld4(a.a_shift + scol);
  } else if (LOADER == BNBWD) {
    k_a = ld4(a.bn5 + scol); k_b = ld4(a.bn5 + a.K + scol); k_c = ld4(a.bn5 + 2 * a.K + scol);
    k_d = ld4(a.bn5 + 3 * a.K + scol); k_e = ld4(a.bn5 + 4 * a.K + scol);
  }
  // A row-major weight has every lane of a fragment load on a different row (32 cache lines per instruction: the forward
  // product ran 11 us where the K-major dX product of the same size ran 6).  With a.wlds (bf16 products, row-major weight)
  // both operands go through LDS as bf16: the A tile [32][K + 4] and, 128 columns of K at a time, the workgroup's 128 x 128
  // weight chunk [128][WKC + 4] — coalesced row segments from global memory, fragments by ds_read_b64, no conversions in
  // the product loop.
  constexpr int WKC = 128;
  const int ldb = a.K + 4;   // shorts per A row (wlds)
  const int wld = WKC + 4;   // shorts per staged weight row
  short *sAb = reinterpret_cast<short *>(sA);
  short *sW = sAb + 32 * ldb;
  double s1 = 0.0, s2 = 0.0;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long row0 = tile * 32;
    WChunk wa, wb;
    if (active && !a.wlds) w_load(a, col, 0, half, wa);
    // ---- stage the A tile: 8 chunks (+ 8 of G) per thread in flight, then transform and write to LDS
    for (int e0 = threadIdx.x; e0 < nch; e0 += 256 * 8) {
      float4 x[8], g[8];
      int srow[8], scl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = min(e0 + 256 * j, nch - 1);  // clamped: unconditional loads, all in flight together
        srow[j] = e / kq;
        scl[j] = (e - srow[j] * kq) * 4;
        x[j] = ld4(a.X + (row0 + srow[j]) * a.ldx + scl[j]);
        if (LOADER == BNBWD) g[j] = ld4(a.G + (row0 + srow[j]) * a.ldx + scl[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (e0 + 256 * j >= nch) break;
        float4 v = x[j];
        if (LOADER == BNRELU) {
          v = make_float4(fmaxf(0.f, v.x * k_a.x + k_b.x), fmaxf(0.f, v.y * k_a.y + k_b.y), fmaxf(0.f, v.z * k_a.z + k_b.z),
                          fmaxf(0.f, v.w * k_a.w + k_b.w));
        } else if (LOADER == BNBWD) {
          const float4 y = x[j], gg = g[j];
          v = make_float4(k_c.x * (gg.x - k_d.x - (y.x * k_a.x + k_b.x) * k_e.x), k_c.y * (gg.y - k_d.y - (y.y * k_a.y + k_b.y) * k_e.y),
                          k_c.z * (gg.z - k_d.z - (y.z * k_a.z + k_b.z) * k_e.z), k_c.w * (gg.w - k_d.w - (y.w * k_a.w + k_b.w) * k_e.w));
        }
        if (a.wlds) {
          short4 pk;
          pk.x = bf16_bits_(v.x); pk.y = bf16_bits_(v.y); pk.z = bf16_bits_(v.z); pk.w = bf16_bits_(v.w);
          *reinterpret_cast<short4 *>(sAb + srow[j] * ldb + scl[j]) = pk;
        } else {
          *reinterpret_cast<float4 *>(sA + srow[j] * lds_ld + scl[j]) = v;
        }
      }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (a.wlds) {  // workgroup-uniform: every wave takes part in staging the weight chunks
      const short *sa = sAb + r * ldb + 4 * half;
      const short *sw = sW + (32 * wave + r) * wld + 4 * half;
      for (int k0 = 0; k0 < a.K; k0 += WKC) {
        const int kc = min(WKC, a.K - k0), k4n = kc / 4;
        if (k0 > 0) __syncthreads();  // the previous chunk's fragment reads are done
        for (int c = threadIdx.x; c < 128 * k4n; c += 256) {
          const int n = c / k4n, k4 = c - n * k4n;
          const int gcol = blockIdx.y * 128 + n;
          const float4 wv = gcol < a.N ? ld4(a.W + (long long)gcol * a.ldw + k0 + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
          short4 pk;
          pk.x = bf16_bits_(wv.x); pk.y = bf16_bits_(wv.y); pk.z = bf16_bits_(wv.z); pk.w = bf16_bits_(wv.w);
          *reinterpret_cast<short4 *>(sW + n * wld + 4 * k4) = pk;
        }
        __syncthreads();
        if (active)
          for (int kb = 0; kb < kc; kb += 16) {
            const short4 a0 = *reinterpret_cast<const short4 *>(sa + k0 + kb), a1 = *reinterpret_cast<const short4 *>(sa + k0 + kb + 8);
            const short4 w0 = *reinterpret_cast<const short4 *>(sw + kb), w1 = *reinterpret_cast<const short4 *>(sw + kb + 8);
            bf16x8_t av, bw;
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            bw[0] = w0.x; bw[1] = w0.y; bw[2] = w0.z; bw[3] = w0.w; bw[4] = w1.x; bw[5] = w1.y; bw[6] = w1.z; bw[7] = w1.w;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc, 0, 0, 0);
          }
      }
    } else if (active) {
      const float *sa = sA + r * lds_ld;
      for (int kb = 0; kb < a.K; kb += 64) {
        if (kb + 32 < a.K) w_load(a, col, kb + 32, half, wb);
        w_mma(sa, kb, half, wa, acc, a.bfm);
        if (kb + 32 >= a.K) break;
        if (kb + 64 < a.K) w_load(a, col, kb + 64, half, wa);
        w_mma(sa, kb + 32, half, wb, acc, a.bfm);
      }
    }
    if (active) {
      // acc[i] = element (row row0 + acc_row(i, half), column col)
      if (EPI == STORE) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[i];
          a.Y[(row0 + acc_row(i, half)) * a.ldy + col] = v;
          ps += v;
          pq += v * v;
        }
        s1 += (double)ps;
        s2 += (double)pq;
      } else if (EPI == BIAS) {
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a.Y[(row0 + acc_row(i, half)) * a.ldy + col] = acc[i] + bv;
      } else {  // MASK
        const float sc = a.p_vec[col], sh = a.p_vec[a.N + col], rs = a.p_vec[2 * a.N + col], nm = a.p_vec[3 * a.N + col];
        float ps = 0.f, pq = 0.f;
        float yv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) yv[i] = a.Yprev[(row0 + acc_row(i, half)) * a.ldprev + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float g = (yv[i] * sc + sh > 0.f) ? acc[i] : 0.f;
          a.Y[(row0 + acc_row(i, half)) * a.ldy + col] = g;
          ps += g;
          pq += g * (yv[i] * rs + nm);
        }
        s1 += (double)ps;
        s2 += (double)pq;
      }
    }
    __syncthreads();  // every wave is done with the A tile before the next one is staged
  }
  if ((EPI == STORE || EPI == MASK) && active) {  // this wave's share of the column reductions -> the workgroup's slab
    const double t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
    if (half == 0) {
      a.stats[((size_t)blockIdx.x * 2) * a.N + col] = t1;
      a.stats[((size_t)blockIdx.x * 2 + 1) * a.N + col] = t2;
    }
  }
}

// out = act(Y * scale + shift): the activation of the LAST BatchNorm layer of a stack, 4 columns per thread.
// act = ReLU, or PReLU with the per-channel slope vector when `slope` is given (relation_module.py:47: Conv1d -> BatchNorm1d
// -> PReLU(C)).
__global__ __launch_bounds__(256) void rows_act_kernel(const float *__restrict__ Y, long long n4, int C,
                                                       const float *__restrict__ vec, const float *__restrict__ slope,
                                                       float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const float4 y = ld4(Y + i * 4), sc = ld4(vec + c), sh = ld4(vec + C + c);
  const float4 sl = slope ? ld4(slope + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float v0 = y.x * sc.x + sh.x, v1 = y.y * sc.y + sh.y, v2 = y.z * sc.z + sh.z, v3 = y.w * sc.w + sh.w;
  *reinterpret_cast<float4 *>(out + i * 4) =
      make_float4(v0 > 0.f ? v0 : sl.x * v0, v1 > 0.f ? v1 : sl.y * v1, v2 > 0.f ? v2 : sl.z * v2, v3 > 0.f ? v3 : sl.w * v3);
}

// G = dOut * act'(Y*scale + shift), per-workgroup column sums of G and G*yhat -> slab [blockIdx.x][2][C]; with a PReLU slope
// also sum(dOut * min(v, 0)) -> dslope slab [blockIdx.x][C].  A workgroup owns `rows_per_block` (<= 64) rows; a thread = four
// consecutive columns x every RL-th row of them (RL = 256 / (C/4) row lanes), all of its rows in flight at once, the column
// sums of the row lanes folded through LDS.  (First form: thread = column walking 64 rows four at a time — 16 dependent
// round trips, 128 of 256 threads idle at C = 128: 23 us for a 2048 x 128 matrix.)
__global__ __launch_bounds__(256) void rows_act_bwd_kernel(const float *__restrict__ dOut, const float *__restrict__ Y,
                                                           long long R, int C, const float *__restrict__ vec,
                                                           const float *__restrict__ slope, long long rows_per_block,
                                                           float *__restrict__ G, double *__restrict__ slabs,
                                                           double *__restrict__ dslope) {
  __shared__ double red[3][256][4];
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  const int c4n = C / 4;                       // float4 chunks per row (C % 4 == 0, C <= 1024: host-checked)
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n, RL = 256 / c4n;  // 256 % c4n == 0 (host-checked)
  const int c = 4 * cl;
  const float4 sc = ld4(vec + c), sh = ld4(vec + C + c), rs = ld4(vec + 2 * C + c), nm = ld4(vec + 3 * C + c);
  const float4 al = slope ? ld4(slope + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0}, s3[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int INF = 8;  // rows in flight per thread
  for (long long rb = r0 + rl; rb < r1; rb += (long long)RL * INF) {
    float4 y[INF], d[INF];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      const long long r = min(rb + (long long)u * RL, R - 1);  // clamped: unconditional loads
      y[u] = ld4(Y + r * C + c);
      d[u] = ld4(dOut + r * C + c);
    }
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      const long long r = rb + (long long)u * RL;
      if (r >= r1) break;
      const float yy[4] = {y[u].x, y[u].y, y[u].z, y[u].w}, dd[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
      const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
      const float nmv[4] = {nm.x, nm.y, nm.z, nm.w}, alv[4] = {al.x, al.y, al.z, al.w};
      float g[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = yy[e] * scv[e] + shv[e];
        g[e] = v > 0.f ? dd[e] : alv[e] * dd[e];
        s1[e] += g[e];
        s2[e] += g[e] * (yy[e] * rsv[e] + nmv[e]);
        s3[e] += v > 0.f ? 0.f : dd[e] * v;
      }
      *reinterpret_cast<float4 *>(G + r * C + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
    red[2][threadIdx.x][e] = s3[e];
  }
  __syncthreads();
  // thread t < C sums column t over the row lanes (fixed order: deterministic)
  for (int col = threadIdx.x; col < C; col += 256) {
    const int cc = col >> 2, e = col & 3;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int k = 0; k < RL; ++k) {
      a1 += red[0][k * c4n + cc][e];
      a2 += red[1][k * c4n + cc][e];
      a3 += red[2][k * c4n + cc][e];
    }
    slabs[((size_t)blockIdx.x * 2) * C + col] = a1;
    slabs[((size_t)blockIdx.x * 2 + 1) * C + col] = a2;
    if (dslope) dslope[(size_t)blockIdx.x * C + col] = a3;
  }
}

// ---- feature propagation glue on point-major rows (pointnet2_modules.py:393-411) -------------------------------------
// X[b,n,:] = [ sum_k w[b,n,k] * known[b, idx[b,n,k], :]  |  unknown[b,n,:] ]   (three_interpolate + torch.cat)
__global__ __launch_bounds__(256) void fp_rows_kernel(const float *__restrict__ known, const float *__restrict__ unknown,
                                                      const int *__restrict__ idx, const float *__restrict__ w, int B,
                                                      int n, int m, int C1, int C2, float *__restrict__ X) {
  const int ld = C1 + C2, q = ld / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * n * q) return;
  const long long bn = t / q;
  const int c = (int)(t - bn * q) * 4;
  float4 v;
  if (c < C1) {
    const int b = (int)(bn / n);
    const int *ip = idx + bn * 3;
    const float *wp = w + bn * 3;
    const float4 p1 = ld4(known + ((long long)b * m + ip[0]) * C1 + c), p2 = ld4(known + ((long long)b * m + ip[1]) * C1 + c),
                 p3 = ld4(known + ((long long)b * m + ip[2]) * C1 + c);
    const float w1 = wp[0], w2 = wp[1], w3 = wp[2];
    v = make_float4(vlp3d_blend3(p1.x, w1, p2.x, w2, p3.x, w3), vlp3d_blend3(p1.y, w1, p2.y, w2, p3.y, w3),
                    vlp3d_blend3(p1.z, w1, p2.z, w2, p3.z, w3), vlp3d_blend3(p1.w, w1, p2.w, w2, p3.w, w3));
  } else {
    v = ld4(unknown + bn * C2 + (c - C1));
  }
  *reinterpret_cast<float4 *>(X + bn * ld + c) = v;
}

// adjoint: d_known[b, j, :] += w * dX[b, n, :C1] over the three neighbours (m known points of a scene accumulate in LDS:
// one workgroup = one scene x a slab of CH channels), d_unknown = dX[:, C1:] is a VIEW taken by the caller.
template <int CH>
__global__ __launch_bounds__(256) void fp_rows_grad_kernel(const float *__restrict__ dX, const int *__restrict__ idx,
                                                           const float *__restrict__ w, int n, int m, int C1, int ld,
                                                           float *__restrict__ dknown) {
  extern __shared__ float acc[];  // [m][CH]
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  for (int i = threadIdx.x; i < m * CH; i += 256) acc[i] = 0.f;
  __syncthreads();
  for (int e = threadIdx.x; e < n * CH; e += 256) {
    const int p = e / CH, c = e - p * CH;
    const long long bn = (long long)b * n + p;
    const float g = dX[bn * ld + c0 + c];
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(&acc[idx[bn * 3 + k] * CH + c], g * w[bn * 3 + k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m * CH; i += 256) {
    const int j = i / CH, c = i - j * CH;
    dknown[((long long)b * m + j) * C1 + c0 + c] = acc[i];
  }
}

unsigned rows_blocks(long long R) {
  const long long t = R / 32;
  return (unsigned)(t < 1024 ? t : 1024);
}

template <int LOADER, int EPI>
int launch(const RowsArgs &a, hipStream_t s) {
  const dim3 grid(rows_blocks(a.R), (a.N + 127) / 128);
  size_t lds = (size_t)32 * (a.K + 4) * sizeof(float);
  RowsArgs b = a;
  b.wlds = a.bfm && !a.wt && a.K % 16 == 0;
  if (b.wlds) lds = ((size_t)32 * (a.K + 4) + (size_t)128 * (128 + 4)) * sizeof(short);  // bf16 A tile + one weight chunk
  auto kern = rows_gemm_kernel<LOADER, EPI>;
  if (lds > 64 * 1024) {
    if (lds > 150 * 1024) return VLP3D_EINVAL;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, b);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

bool bad_gemm(const float *X, const float *W, const float *Y, long long R, int K, int N, int ldx, int ldw, int ldy) {
  return !X || !W || !Y || R < 32 || (R & 31) || R > (1ll << 24) || K < 32 || (K & 31) || K > 1024 || N < 32 || (N & 31) || ldx < K ||
         (ldx & 3) || ldw < 1 || ldy < N;
}

// The same adjoint through the INVERSE of the three_nn map (vlp3d_sa_inverse(idx as (B, n, 3), N = m): inv_start (B*m + 1),
// inv_refs = flat (b*n + p)*3 + k): a wave per known point sums w[ref] * dX[ref / 3][:C1] over its references with 16-byte
// loads down whole rows — no LDS atomics, every output written once.  (The LDS-atomic form reads dX in 32-byte pieces, one
// workgroup per scene and 8-channel slab: 34 + 17 us for the two FP modules of cfg2.)
__global__ __launch_bounds__(256) void fp_rows_grad_csr_kernel(const float *__restrict__ dX, const float *__restrict__ w,
                                                               const int *__restrict__ inv_start, const int *__restrict__ inv_refs,
                                                               int nknown, int C1, int ld, float *__restrict__ dknown) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= nknown) return;
  const int s0 = inv_start[j], cnt = inv_start[j + 1] - s0;
  for (int c = 4 * lane; c < C1; c += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {  // four references in flight
      int ref[4];
      float wt[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ref[u] = inv_refs[s0 + i + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wt[u] = w[ref[u]];
        v[u] = ld4(dX + (long long)(ref[u] / 3) * ld + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x += wt[u] * v[u].x; acc.y += wt[u] * v[u].y; acc.z += wt[u] * v[u].z; acc.w += wt[u] * v[u].w;
      }
    }
    for (; i < cnt; ++i) {
      const int ref = inv_refs[s0 + i];
      const float wt = w[ref];
      const float4 v = ld4(dX + (long long)(ref / 3) * ld + c);
      acc.x += wt * v.x; acc.y += wt * v.y; acc.z += wt * v.z; acc.w += wt * v.w;
    }
    *reinterpret_cast<float4 *>(dknown + (long long)j * C1 + c) = acc;
  }
}

}  // namespace

// workgroups (= statistic slabs of 2*N doubles each) a rows product over R rows is launched with
extern "C" int vlp3d_rows_slabs(long long R) { return R < 32 ? 0 : (int)rows_blocks(R); }

// Y (R x N, row stride ldy) = A W^T [+ bias], A = X (a_vec NULL) or relu(X*scale + shift) (a_vec = [scale | shift | ..],
// each of length K: the `vec` of vlp3d_sa_bn_fold).  stats != NULL: no bias, per-column sums of Y -> slabs
// [vlp3d_rows_slabs(R)][2][N] for vlp3d_sa_bn_fold.  W (N x K) row-major.  R % 32 == 0, K % 32 == 0, N % 32 == 0.
extern "C" int vlp3d_rows_fwd(const float *X, int ldx, long long R, int K, const float *a_vec, const float *W,
                              const float *bias, int N, float *Y, int ldy, double *stats, int bf16_mma, void *stream) {
  if (bad_gemm(X, W, Y, R, K, N, ldx, K, ldy) || (stats && bias) || (a_vec && (256 % (K / 4)))) return VLP3D_EINVAL;
  RowsArgs a = {};
  a.bfm = bf16_mma != 0;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = K; a.K = K; a.N = N; a.R = R; a.Y = Y; a.ldy = ldy; a.bias = bias; a.stats = stats;
  if (a_vec) { a.a_scale = a_vec; a.a_shift = a_vec + K; }
  hipStream_t s = (hipStream_t)stream;
  if (a_vec) return stats ? launch<BNRELU, STORE>(a, s) : launch<BNRELU, BIAS>(a, s);
  return stats ? launch<PLAIN, STORE>(a, s) : launch<PLAIN, BIAS>(a, s);
}

// The same product with the weight given K-major: WT (K x ldw floats), WT[k][n] = W[n][k] (vlp3d_transpose_batch writes
// such copies of all the stacks' weights in one launch per step).  The fragment loads are then coalesced over the lanes'
// 32 output columns — the path the input-gradient product has always used — instead of re-staging the workgroup's
// 128 x K weight block through LDS for every 32-row tile: 30 -> 13 us for 8192 x 256 x 256 (round 3).
extern "C" int vlp3d_rows_fwd_wt(const float *X, int ldx, long long R, int K, const float *a_vec, const float *WT, int ldw,
                                 const float *bias, int N, float *Y, int ldy, double *stats, int bf16_mma, void *stream) {
  if (bad_gemm(X, WT, Y, R, K, N, ldx, ldw, ldy) || ldw < N || (stats && bias) || (a_vec && (256 % (K / 4)))) return VLP3D_EINVAL;
  RowsArgs a = {};
  a.bfm = bf16_mma != 0;
  a.X = X; a.ldx = ldx; a.W = WT; a.ldw = ldw; a.wt = 1; a.K = K; a.N = N; a.R = R; a.Y = Y; a.ldy = ldy; a.bias = bias;
  a.stats = stats;
  if (a_vec) { a.a_scale = a_vec; a.a_shift = a_vec + K; }
  hipStream_t s = (hipStream_t)stream;
  if (a_vec) return stats ? launch<BNRELU, STORE>(a, s) : launch<BNRELU, BIAS>(a, s);
  return stats ? launch<PLAIN, STORE>(a, s) : launch<PLAIN, BIAS>(a, s);
}

// Input gradient of a layer Y = A W^T with W (N x K): dA (R x K) = dY W, dY = G (bn5 NULL) or BatchNorm-backward of
// (G, Ypre) with the constants bn5 [5][N] of vlp3d_sa_bn_bwd_consts.  p_vec != NULL: A was relu(BN(Yprev)) — the result
// is masked with relu'(Yprev*scale + shift) (p_vec = that layer's `vec` [4][K]) and the column sums of the masked
// gradient / gradient * yhat_prev go to tstats slabs [vlp3d_rows_slabs(R)][2][K]; p_vec NULL: plain store (the stack's
// input gradient).  K % 32 == 0, N % 32 == 0.
extern "C" int vlp3d_rows_dgrad(const float *G, const float *Ypre, int ldg, const float *bn5, const float *W, long long R,
                                int N, int K, const float *Yprev, int ldprev, const float *p_vec, float *dA, int lda,
                                double *tstats, int bf16_mma, void *stream) {
  if (bad_gemm(G, W, dA, R, N, K, ldg, K, lda) || (bn5 && (!Ypre || (256 % (N / 4)))) ||
      (p_vec && (!Yprev || !tstats || ldprev < K)))
    return VLP3D_EINVAL;
  RowsArgs a = {};
  a.bfm = bf16_mma != 0;
  a.ldx = ldg; a.W = W; a.ldw = K; a.wt = 1; a.K = N; a.N = K; a.R = R; a.Y = dA; a.ldy = lda;
  if (bn5) { a.X = Ypre; a.G = G; a.bn5 = bn5; } else { a.X = G; }
  a.Yprev = Yprev; a.ldprev = ldprev; a.p_vec = p_vec; a.stats = tstats;
  hipStream_t s = (hipStream_t)stream;
  if (bn5) return p_vec ? launch<BNBWD, MASK>(a, s) : launch<BNBWD, BIAS>(a, s);
  return p_vec ? launch<PLAIN, MASK>(a, s) : launch<PLAIN, BIAS>(a, s);
}

// out (R x C) = relu(Y * scale + shift), vec = [scale | shift | ..] of length C each.  C % 4 == 0.
extern "C" int vlp3d_rows_act(const float *Y, long long R, int C, const float *vec, const float *slope, float *out,
                              void *stream) {
  if (!Y || !vec || !out || R < 1 || C < 4 || (C & 3)) return VLP3D_EINVAL;
  const long long n4 = R * C / 4;
  hipLaunchKernelGGL(rows_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Y, n4, C, vec, slope,
                     out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// G (R x C) = dOut masked by the ReLU of the last BatchNorm layer, tstats slabs [nslab][2][C] (nslab returned by
// vlp3d_rows_act_slabs) for vlp3d_sa_bn_bwd_consts.  vec = that layer's [scale | shift | rstd | -mean*rstd].
extern "C" int vlp3d_rows_act_slabs(long long R) {
  const long long n = (R + 31) / 32;  // 32 rows per workgroup up to 256 workgroups (round 3: 64 -> 32, one per CU at 8192 rows)
  return (int)(n < 256 ? n : 256);
}
extern "C" int vlp3d_rows_act_bwd(const float *dOut, const float *Y, long long R, int C, const float *vec, const float *slope,
                                  float *G, double *tstats, double *dslope_slabs, void *stream) {
  // C / 4 must divide 256 (a thread keeps its four columns): 64, 128, 256, 512, 1024 — the widths the BatchNorm loaders of the
  // rows products accept as well (row_mlp.supported)
  if (!dOut || !Y || !vec || !G || !tstats || R < 1 || C < 4 || (C & 3) || C > 1024 || (256 % (C / 4)) || (slope && !dslope_slabs))
    return VLP3D_EINVAL;
  const int nslab = vlp3d_rows_act_slabs(R);
  const long long rpb = (R + nslab - 1) / nslab;
  hipLaunchKernelGGL(rows_act_bwd_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, dOut, Y, R, C, vec, slope, rpb, G, tstats,
                     slope ? dslope_slabs : nullptr);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// d_known (B*m, C1) through the inverse three_nn map (see fp_rows_grad_csr_kernel); C1 % 4 == 0, ld >= C1 and % 4 == 0.
extern "C" int vlp3d_fp_rows_grad_csr(const float *dX, const float *weight, const int *inv_start, const int *inv_refs, int B,
                                      int m, int C1, int ld, float *d_known, void *stream) {
  if (!dX || !weight || !inv_start || !inv_refs || !d_known || B < 1 || m < 1 || C1 < 4 || (C1 & 3) || ld < C1 || (ld & 3))
    return VLP3D_EINVAL;
  const int nk = B * m;
  hipLaunchKernelGGL(fp_rows_grad_csr_kernel, dim3((nk + 3) / 4), dim3(256), 0, (hipStream_t)stream, dX, weight, inv_start,
                     inv_refs, nk, C1, ld, d_known);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// Feature-propagation rows: X (B*n, C1 + C2) = [three_interpolate(known (B,m,C1) point-major, idx, weight) | unknown
// (B,n,C2) point-major] — pointnet2_modules.py:393-411 without the (B,C,n) round trips.  C1, C2 % 4 == 0.
extern "C" int vlp3d_fp_rows(const float *known, const float *unknown, const int *idx, const float *weight, int B, int n,
                             int m, int C1, int C2, float *X, void *stream) {
  if (!known || !unknown || !idx || !weight || !X || B < 1 || n < 1 || m < 1 || C1 < 4 || (C1 & 3) || C2 < 4 || (C2 & 3))
    return VLP3D_EINVAL;
  const long long total = (long long)B * n * ((C1 + C2) / 4);
  hipLaunchKernelGGL(fp_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, known,
                     unknown, idx, weight, B, n, m, C1, C2, X);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// d_known (B,m,C1) = adjoint of the interpolation part of vlp3d_fp_rows applied to dX[:, :C1] (row stride ld), fully
// written.  m * 16 floats of LDS per workgroup: m <= 1024.  C1 % 16 == 0.
extern "C" int vlp3d_fp_rows_grad(const float *dX, const int *idx, const float *weight, int B, int n, int m, int C1, int ld,
                                  float *d_known, void *stream) {
  if (!dX || !idx || !weight || !d_known || B < 1 || n < 1 || m < 1 || m > 1024 || C1 < 16 || (C1 & 15) || ld < C1)
    return VLP3D_EINVAL;
  static const int ch = getenv("VLP3D_FPGRAD_CH") ? atoi(getenv("VLP3D_FPGRAD_CH")) : 8;  // 16: only C1/16 x B = 128 workgroups
  if (ch == 8)
    hipLaunchKernelGGL(fp_rows_grad_kernel<8>, dim3(C1 / 8, B), dim3(256), (size_t)m * 8 * sizeof(float),
                       (hipStream_t)stream, dX, idx, weight, n, m, C1, ld, d_known);
  else if (ch == 4)
    hipLaunchKernelGGL(fp_rows_grad_kernel<4>, dim3(C1 / 4, B), dim3(256), (size_t)m * 4 * sizeof(float),
                       (hipStream_t)stream, dX, idx, weight, n, m, C1, ld, d_known);
  else
    hipLaunchKernelGGL(fp_rows_grad_kernel<16>, dim3(C1 / 16, B), dim3(256), (size_t)m * 16 * sizeof(float),
                       (hipStream_t)stream, dX, idx, weight, n, m, C1, ld, d_known);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
