// nn.Linear forward / input gradient for MANY rows (R >= 8192: the decoder layers and the match MLP work on
// B*L*K = 16 384 rows) with bf16 MFMA operands — the timing configuration of csrc/sa_mlp.hip's linear_bf16_kernel.
//
// linear_bf16_kernel gives every 32 x 32 output tile its own wave, whose lanes read "their" operand rows straight from
// memory: 32 rows x 16 bytes per load instruction (32 cache lines), and the A rows are read again by every column wave —
// 12 us for 16384 x 128 x 128 (16.8 MB: 1.4 TB/s).  Here a workgroup owns a 64-row x 128-column output block:
//   * the A block [64][KC] and the weight block [128][KC] go through LDS as bf16, staged with coalesced 16-byte row
//     segments (each A element is fetched ONCE per 128 output columns, each weight element once per workgroup and chunk);
//   * four waves = 2 row tiles x 2 column pairs, two 32 x 32 accumulators each; fragments are one ds_read_b128 each
//     (row stride KC + 8 shorts = 272 B: conflict-free);
//   * the reduction dimension runs in chunks of KC = 128, so any K (forward) / N (input gradient) that is a multiple of
//     16 works with 51 KB of LDS (three workgroups per CU);
//   * WT: the weight is used as stored (dX = dY W, W (N x K) row-major): the chunk keeps its [k][column] image and a
//     fragment is read as eight 2-byte column reads.
// Epilogue: + bias (forward) or + base (the gradient arriving through a residual connection), fp32 stores with the
// column on the lane (128-byte row segments).
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 128;       // reduction chunk
constexpr int LD = KC + 8;    // LDS row stride (shorts)
constexpr int TC = 128;       // columns per workgroup block

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

// Y (R x ncols) = X (R x kdim) * op(W) [+ bias] [+ base]
//   WT = false: op(W) = W^T, W (ncols x kdim) row-major with row stride ldw        (forward: ncols = N, kdim = K)
//   WT = true : op(W) = W,   W (kdim x ncols) row-major with row stride ldw        (input gradient: kdim = N, ncols = K)
// TR = rows per workgroup block: 64 (waves 2 x 2, two accumulators each) or 32 (waves 1 x 4, one accumulator each: twice the
// workgroups — two per CU at 16 384 rows — for the same LDS weight traffic per workgroup)
// WB (with WT = false only): W already holds bf16 (a prepared weight of a grouped-MLP stack): staged without conversion.
// YB (forward only): Y receives bf16 rows (row stride ldy elements) — a projection an attention core reads next.
template <bool WT, int TR, bool WB = false, bool YB = false>
__global__ __launch_bounds__(256) void linear_tile_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw,
                                                          int kdim, int ncols, const float *__restrict__ bias,
                                                          const float *__restrict__ base, long long R, float *__restrict__ Y, int ldy) {
  __shared__ __attribute__((aligned(16))) short sA[TR * LD];
  __shared__ __attribute__((aligned(16))) short sW[TC * LD];
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NT = TR / 32;                      // accumulators per wave
  const int wr = TR == 64 ? (wave & 1) : 0;        // row tile of this wave
  const int wc = TR == 64 ? (wave >> 1) : wave;    // column group: 64 columns (TR = 64) or 32 (TR = 32)
  constexpr int CW = 32 * NT;                      // columns per wave
  const int c0 = blockIdx.y * TC;
  const long long nblk = (R + TR - 1) / TR;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long row0 = blk * TR;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int k0 = 0; k0 < kdim; k0 += KC) {
      const int kc = min(KC, kdim - k0);     // multiple of 16
      const int q = kc / 4;                  // float4 chunks per row
      __syncthreads();                       // the previous chunk's fragment reads are done
      if (kc == KC) {
        // full chunk: fixed trip counts, ALL 24 loads of a thread in flight before the first conversion (the generic loop
        // below spends an integer division per element and lets the compiler serialise load -> convert -> store)
        float4 va[TR * (KC / 4) / 256], vw[TC * (KC / 4) / 256];
#pragma unroll
        for (int j = 0; j < TR * (KC / 4) / 256; ++j) {
          const int e = threadIdx.x + 256 * j, row = e >> 5, c4 = e & 31;
          const long long rr = min(row0 + row, R - 1);  // clamped: unconditional loads (rows past R are never stored)
          va[j] = ld4(X + rr * ldx + k0 + 4 * c4);
        }
        uint2 vb[WB ? TC * (KC / 4) / 256 : 1];
        if (WB) {
          const short *Wb = reinterpret_cast<const short *>(W);
#pragma unroll
          for (int j = 0; j < TC * (KC / 4) / 256; ++j) {
            const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
            vb[j] = *reinterpret_cast<const uint2 *>(Wb + (long long)min(c0 + col, ncols - 1) * ldw + k0 + 4 * c4);
          }
        } else if (!WT) {
#pragma unroll
          for (int j = 0; j < TC * (KC / 4) / 256; ++j) {
            const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
            vw[j] = ld4(W + (long long)min(c0 + col, ncols - 1) * ldw + k0 + 4 * c4);
          }
        } else {
#pragma unroll
          for (int j = 0; j < TC * (KC / 4) / 256; ++j) {
            const int e = threadIdx.x + 256 * j, k = e >> 5, c4 = e & 31;
            vw[j] = ld4(W + (long long)(k0 + k) * ldw + min(c0 + 4 * c4, ncols - 4));
          }
        }
#pragma unroll
        for (int j = 0; j < TR * (KC / 4) / 256; ++j) {
          const int e = threadIdx.x + 256 * j, row = e >> 5, c4 = e & 31;
          *reinterpret_cast<bf16x4 *>(sA + row * LD + 4 * c4) = pack4(va[j]);
        }
        if (WB) {
#pragma unroll
          for (int j = 0; j < TC * (KC / 4) / 256; ++j) {
            const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
            *reinterpret_cast<uint2 *>(sW + col * LD + 4 * c4) = vb[j];
          }
        } else if (!WT) {
#pragma unroll
          for (int j = 0; j < TC * (KC / 4) / 256; ++j) {
            const int e = threadIdx.x + 256 * j, col = e >> 5, c4 = e & 31;
            *reinterpret_cast<bf16x4 *>(sW + col * LD + 4 * c4) = pack4(vw[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < TC * (KC / 4) / 256; ++j) {
            const int e = threadIdx.x + 256 * j, k = e >> 5, c4 = e & 31;
            *reinterpret_cast<bf16x4 *>(sW + k * LD + 4 * c4) = pack4(vw[j]);   // natural [k][column] image (TC == KC)
          }
        }
      } else {
      // ---- A block: TR rows x kc columns
      for (int e = threadIdx.x; e < TR * q; e += 256) {
        const int row = e / q, c4 = e - row * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < R) v = ld4(X + (row0 + row) * ldx + k0 + 4 * c4);
        *reinterpret_cast<bf16x4 *>(sA + row * LD + 4 * c4) = pack4(v);
      }
      // ---- weight block: [column][k]
      if (WB) {
        const short *Wb = reinterpret_cast<const short *>(W);
        for (int e = threadIdx.x; e < TC * q; e += 256) {
          const int col = e / q, c4 = e - col * q;
          uint2 v = make_uint2(0u, 0u);
          if (c0 + col < ncols) v = *reinterpret_cast<const uint2 *>(Wb + (long long)(c0 + col) * ldw + k0 + 4 * c4);
          *reinterpret_cast<uint2 *>(sW + col * LD + 4 * c4) = v;
        }
      } else if (!WT) {
        for (int e = threadIdx.x; e < TC * q; e += 256) {
          const int col = e / q, c4 = e - col * q;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c0 + col < ncols) v = ld4(W + (long long)(c0 + col) * ldw + k0 + 4 * c4);
          *reinterpret_cast<bf16x4 *>(sW + col * LD + 4 * c4) = pack4(v);
        }
      } else {  // W rows are the reduction index: read row segments (columns contiguous), write transposed
        for (int e = threadIdx.x; e < kc * (TC / 4); e += 256) {
          const int k = e / (TC / 4), c4 = e - k * (TC / 4);
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c0 + 4 * c4 < ncols) v = ld4(W + (long long)(k0 + k) * ldw + c0 + 4 * c4);
          *reinterpret_cast<bf16x4 *>(sW + k * LD + 4 * c4) = pack4(v);
        }
      }
      }
      __syncthreads();
      const short *pa = sA + (32 * wr + r) * LD + 8 * half;
      if (!WT) {
        const short *pw = sW + (CW * wc + r) * LD + 8 * half;
        for (int s = 0; s < kc / 16; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8 *>(pa + 16 * s);
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *reinterpret_cast<const bf16x8 *>(pw + 32 * t * LD + 16 * s), acc[t], 0, 0, 0);
        }
      } else {
        // the weight chunk sits in its natural [k][column] image (coalesced, conflict-free staging; a transposed image
        // made the staging writes 16-way bank conflicted): a fragment = eight 2-byte reads down a column, lanes side by side
        const short *pw = sW + (8 * half) * LD + CW * wc + r;
        for (int s = 0; s < kc / 16; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8 *>(pa + 16 * s);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = pw[(16 * s + j) * LD + 32 * t];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = c0 + CW * wc + 32 * t + r;
      if (col >= ncols) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long long row = row0 + 32 * wr + acc_row(i, half);
        if (row < R) {
          const long long o = row * ldy + col;
          if (YB) reinterpret_cast<short *>(Y)[o] = bf16_bits(acc[t][i] + bv);
          else Y[o] = acc[t][i] + bv + (base ? base[o] : 0.f);
        }
      }
    }
  }
}

}  // namespace

// Internal launcher (not part of include/vlp3d.h): called by vlp3d_linear_fwd / vlp3d_linear_dgrad of csrc/sa_mlp.hip for the
// many-row shapes.  transposed_weight 0: Y = X W^T (+ bias); 1: Y = X W (+ base).  kdim % 16 == 0, ncols % 4 == 0.
int vlp3d_internal_linear_tile(const float *X, int ldx, const float *W, int ldw, int kdim, int ncols, const float *bias,
                               const float *base, long long R, float *Y, int ldy, int transposed_weight, hipStream_t stream) {
  if (kdim % 16 || ncols % 4 || R < 1) return -22;
  static const int tr = getenv("VLP3D_LINEAR_TILE_ROWS") ? atoi(getenv("VLP3D_LINEAR_TILE_ROWS")) : 32;  // (round 4 sweep: 64 -> 32)
  const int TRr = tr == 32 ? 32 : 64;
  const long long nblk = (R + TRr - 1) / TRr;
  const dim3 grid((unsigned)(nblk < 4096 ? nblk : 4096), (unsigned)((ncols + TC - 1) / TC));
#define VLP3D_LT(WTv, TRv) hipLaunchKernelGGL((linear_tile_kernel<WTv, TRv>), grid, dim3(256), 0, stream, X, ldx, W, ldw, kdim, ncols, bias, base, R, Y, ldy)
  if (transposed_weight) { if (TRr == 32) VLP3D_LT(true, 32); else VLP3D_LT(true, 64); }
  else { if (TRr == 32) VLP3D_LT(false, 32); else VLP3D_LT(false, 64); }
#undef VLP3D_LT
  VLP3D_LAUNCH_CHECK();
  return 0;
}

// Y (R x ncols) = X (R x kdim) W^T with W (ncols x kdim) ALREADY bf16 (row stride ldw elements): csrc/sa_gather_sum.hip.
int vlp3d_internal_linear_tile_w16(const float *X, int ldx, const void *Wbf16, int ldw, int kdim, int ncols, long long R,
                                   float *Y, int ldy, hipStream_t stream) {
  if (kdim % 16 || ncols % 4 || ldw % 4 || R < 1) return -22;
  const long long nblk = (R + 63) / 64;
  const dim3 grid((unsigned)(nblk < 4096 ? nblk : 4096), (unsigned)((ncols + TC - 1) / TC));
  hipLaunchKernelGGL((linear_tile_kernel<false, 64, true>), grid, dim3(256), 0, stream, X, ldx, (const float *)Wbf16, ldw, kdim, ncols,
                     (const float *)nullptr, (const float *)nullptr, R, Y, ldy);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

// Y16 (R x N bf16 rows) = X (R x K fp32) W^T + bias with bf16 MFMA operands: vlp3d_linear_fwd(bf16_mma = 1) whose result is
// stored as bf16 — the query projection in front of an attention core that takes bf16 rows (vlp3d_sdpa_fwd_io, io bit 1).
// R % 32 == 0, K % 16 == 0, N % 64 == 0.
extern "C" int vlp3d_linear_fwd_rows16(const float *X, const float *W, const float *bias, long long R, int K, int N, void *Y16,
                                       void *stream) {
  if (!X || !W || !Y16 || R < 32 || (R & 31) || K < 16 || (K & 15) || N < 64 || (N & 63)) return VLP3D_EINVAL;
  const long long nblk = (R + 31) / 32;
  const dim3 grid((unsigned)(nblk < 4096 ? nblk : 4096), (unsigned)((N + TC - 1) / TC));
  hipLaunchKernelGGL((linear_tile_kernel<false, 32, false, true>), grid, dim3(256), 0, (hipStream_t)stream, X, K, W, K, K, N, bias,
                     (const float *)nullptr, R, reinterpret_cast<float *>(Y16), N);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
