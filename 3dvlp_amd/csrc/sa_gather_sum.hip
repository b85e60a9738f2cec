// Backward of a grouped MLP's gather layer without atomics (bf16 configuration, backbone levels sa2..sa4).
//
// Layer 1 of a PointNet++ SA module is A0 = [features[idx] | local xyz] followed by the 1x1 conv: dA0 = dY1 W1, and the
// feature part of dA0 is scattered back onto the source points.  Gather and conv commute with that sum:
//     d(features)[p] = sum_{rows r that gathered p} (dY1[r] W1[:, features]) = (sum_r dY1[r]) W1[:, features]
// so instead of a (rows x 259) product whose every element becomes a float atomic (csrc/sa_mlp.hip row_gemm<BNBWD, SCATTER>:
// 48 / 89 / 50 us for sa2 / sa3 / sa4 at cfg2, 12 M atomics at sa3), ONE pass sums the 64 / 128-wide dY1 rows of every point
// through the inverse map of vlp3d_sa_inverse (a wave per point, 16-byte loads, no atomics, every output written once), and
// ONE (B*N) x C0 x C product on the matrix cores (csrc/linear_tile.hip, weight = the stack's prepared bf16 W1^T) gives
// d(features).  dY1 = BatchNorm backward of (G1, Y1): k1 (G - w (k2 + yhat k3)) with the row multiplicity w of the compact
// map, exactly the loader of the kernel it replaces.  Coordinates take no gradient at these levels (the geometry of the
// backbone depends on the input cloud only); the vote aggregation, whose coordinates are learned, keeps the scatter form.
#include <hip/hip_bf16.h>

#include "common.h"

int vlp3d_internal_linear_tile_w16(const float *X, int ldx, const void *Wbf16, int ldw, int kdim, int ncols, long long R,
                                   float *Y, int ldy, hipStream_t stream);

namespace {

__device__ __forceinline__ void unpack8(const uint4 &v, float (&o)[8]) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(w[i] << 16);
    o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

template <int C0>
__global__ __launch_bounds__(256) void sa_gather_sum_kernel(const unsigned short *__restrict__ G, const unsigned short *__restrict__ Y,
                                                            const float *__restrict__ bn5, const int4 *__restrict__ crow,
                                                            const int *__restrict__ inv_start, const int *__restrict__ inv_rows,
                                                            int npoints, float *__restrict__ Gsum) {
  constexpr int LPR = C0 / 8;     // lanes per row (8 channels = 16 bytes of bf16 each)
  constexpr int RPW = 64 / LPR;   // rows in flight per wave
  const int lane = threadIdx.x & 63, sub = lane / LPR, cl = lane % LPR;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= npoints) return;
  float k1[8], ky[8], k0[8];  // dY = k1 g - w (k0 + ky y):  k0 = k1 (k2 + nm k3), ky = k1 k3 rs
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 8 * cl + i;
    const float rs = bn5[c], nm = bn5[C0 + c], a = bn5[2 * C0 + c], b2 = bn5[3 * C0 + c], b3 = bn5[4 * C0 + c];
    k1[i] = a;
    k0[i] = a * (b2 + nm * b3);
    ky[i] = a * b3 * rs;
  }
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const int s0 = inv_start[p], n = inv_start[p + 1] - s0;
  for (int i = sub; i < n; i += RPW) {
    const int r = inv_rows[s0 + i];
    const uint4 gv = *reinterpret_cast<const uint4 *>(G + (long long)r * C0 + 8 * cl);
    const uint4 yv = *reinterpret_cast<const uint4 *>(Y + (long long)r * C0 + 8 * cl);
    const float w = crow ? __int_as_float(crow[r].z) : 1.f;
    float g[8], y[8];
    unpack8(gv, g);
    unpack8(yv, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __builtin_fmaf(k1[e], g[e], -w * __builtin_fmaf(ky[e], y[e], k0[e]));
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], off);
  if (sub == 0) {
    float *o = Gsum + (long long)p * C0 + 8 * cl;
    *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

}  // namespace

// d(features) (B*N, C) of a bf16 grouped MLP's gather layer.  G, Y: (rows, c0) bf16 masked gradient / pre-activation of
// layer 1; bn5 (5 x c0); WT: the stack's prepared W1^T (kpad x c0) bf16, rows = input channels in [features | xyz | 0]
// order; crow: compact map or NULL; inv_start / inv_rows: vlp3d_sa_inverse; Gsum: (B*N, c0) fp32 scratch.  c0 in {64, 128},
// C % 4 == 0.
extern "C" int vlp3d_sa_bwd_gather_csr(const void *G, const void *Y, int c0, const float *bn5, const void *WT, const void *crow,
                                       const int *inv_start, const int *inv_rows, int B, int N, int C, float *Gsum,
                                       float *dfeat_pm, void *stream) {
  if (!G || !Y || !bn5 || !WT || !inv_start || !inv_rows || !Gsum || !dfeat_pm || B < 1 || N < 1 || (c0 != 64 && c0 != 128) ||
      C < 4 || (C & 3))
    return -22;
  hipStream_t s = (hipStream_t)stream;
  const int np = B * N;
  const dim3 grid((np + 3) / 4);
  if (c0 == 64)
    hipLaunchKernelGGL(sa_gather_sum_kernel<64>, grid, dim3(256), 0, s, (const unsigned short *)G, (const unsigned short *)Y, bn5,
                       (const int4 *)crow, inv_start, inv_rows, np, Gsum);
  else
    hipLaunchKernelGGL(sa_gather_sum_kernel<128>, grid, dim3(256), 0, s, (const unsigned short *)G, (const unsigned short *)Y, bn5,
                       (const int4 *)crow, inv_start, inv_rows, np, Gsum);
  VLP3D_LAUNCH_CHECK();
  return vlp3d_internal_linear_tile_w16(Gsum, c0, WT, c0, c0, C, np, dfeat_pm, C, s);
}
