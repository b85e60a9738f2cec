// group_rows: the QueryAndGroup gather in the layout the matrix cores want — replaces
// group_points(xyz) + in-place (-= centre, /= radius) + group_points(features) + cat of
// lib/pointnet2/pointnet2_utils.py:343-355 (four passes over a (B,3+C,M,S) tensor, NCHW) by ONE kernel
// that writes GEMM-ready rows:
//     out[r][0..C)   = feat_pm[b][idx[r]][0..C)                      (features, POINT-MAJOR (B,N,C) source)
//     out[r][C..C+3) = (xyz[b][idx[r]] - new_xyz[b][m]) / radius     (normalised local coordinates; radius=1: off)
//     out[r][C+3]    = 0                                              (pad: row length C+4, 8/16-byte aligned)
// with r = (b*M + m)*S + s.  A neighbour's C channels are contiguous (132..256 floats), so every gather is a
// coalesced 512-1024 byte row read instead of C scattered 4-byte reads, and the adjoint adds whole rows with
// contiguous float atomics (the shape MI355X's memory-side atomics run at full rate for).
// The 1x1-conv weight of the first MLP layer is applied to these rows with its columns permuted
// ([features, xyz, 0] instead of [xyz, features]) — same contraction, different summation order.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

__device__ __forceinline__ void store4(float *dst, float4 v) { *reinterpret_cast<float4 *>(dst) = v; }
__device__ __forceinline__ void store4(__hip_bfloat16 *dst, float4 v) {
  union {
    __hip_bfloat16 h[4];
    uint2 u;
  } p;
  p.h[0] = __float2bfloat16(v.x); p.h[1] = __float2bfloat16(v.y);
  p.h[2] = __float2bfloat16(v.z); p.h[3] = __float2bfloat16(v.w);
  *reinterpret_cast<uint2 *>(dst) = p.u;
}
__device__ __forceinline__ float4 load4(const float *src) { return *reinterpret_cast<const float4 *>(src); }
__device__ __forceinline__ float4 load4(const __hip_bfloat16 *src) {
  union {
    __hip_bfloat16 h[4];
    uint2 u;
  } p;
  p.u = *reinterpret_cast<const uint2 *>(src);
  return make_float4(__bfloat162float(p.h[0]), __bfloat162float(p.h[1]), __bfloat162float(p.h[2]),
                     __bfloat162float(p.h[3]));
}

// one thread = one 4-element chunk of one output row; chunks 0..C/4-1 are features, chunk C/4 is xyz+pad
template <typename OutT>
__global__ __launch_bounds__(256) void group_rows_fwd_kernel(const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             const int *__restrict__ idx,
                                                             const float *__restrict__ feat_pm, int N, int M, int S,
                                                             int C, float radius, OutT *__restrict__ out,
                                                             long long total_chunks) {
  const int cpr = C / 4 + 1;  // chunks per row
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total_chunks; t += (long long)gridDim.x * 256) {
    const long long r = t / cpr;
    const int ch = (int)(t - r * cpr);
    const long long bm = r / S;  // b*M + m
    const long long b = bm / M;
    const int p = idx[r];
    OutT *__restrict__ dst = out + r * (C + 4) + ch * 4;
    if (ch < C / 4) {
      store4(dst, load4(feat_pm + (b * N + p) * C + ch * 4));
    } else {
      const float *__restrict__ q = xyz + (b * N + p) * 3;
      const float *__restrict__ c = new_xyz + bm * 3;
      store4(dst, make_float4((q[0] - c[0]) / radius, (q[1] - c[1]) / radius, (q[2] - c[2]) / radius,
                              0.f));
    }
  }
}

template <typename GT>
__global__ __launch_bounds__(256) void group_rows_bwd_kernel(const GT *__restrict__ dout, const int *__restrict__ idx,
                                                             int N, int M, int S, int C, float radius,
                                                             float *__restrict__ dfeat_pm, float *__restrict__ dxyz,
                                                             float *__restrict__ dnew_xyz, long long total_chunks) {
  const int cpr = C / 4 + 1;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total_chunks; t += (long long)gridDim.x * 256) {
    const long long r = t / cpr;
    const int ch = (int)(t - r * cpr);
    const long long bm = r / S;
    const long long b = bm / M;
    const int p = idx[r];
    const float4 g = load4(dout + r * (C + 4) + ch * 4);
    if (ch < C / 4) {
      if (dfeat_pm != nullptr) {
        float *__restrict__ d = dfeat_pm + (b * N + p) * C + ch * 4;
        atomicAdd(d + 0, g.x); atomicAdd(d + 1, g.y); atomicAdd(d + 2, g.z); atomicAdd(d + 3, g.w);
      }
    } else {
      const float gx = g.x / radius, gy = g.y / radius, gz = g.z / radius;
      if (dxyz != nullptr) {
        float *__restrict__ d = dxyz + (b * N + p) * 3;
        atomicAdd(d + 0, gx); atomicAdd(d + 1, gy); atomicAdd(d + 2, gz);
      }
      if (dnew_xyz != nullptr) {
        float *__restrict__ d = dnew_xyz + bm * 3;
        atomicAdd(d + 0, -gx); atomicAdd(d + 1, -gy); atomicAdd(d + 2, -gz);
      }
    }
  }
}

bool bad(int B, int N, int M, int S, int C) { return B < 1 || N < 1 || M < 1 || S < 1 || C < 4 || (C & 3) != 0; }

unsigned grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  const long long cap = 256 * 32;  // 256 CUs x 32 resident blocks worth of grid-stride
  return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace

extern "C" int vlp3d_group_rows(const float *xyz, const float *new_xyz, const int *idx, const float *feat_pm, int B,
                                int N, int M, int S, int C, float radius, void *out, int out_bf16, void *stream) {
  if (!xyz || !new_xyz || !idx || !feat_pm || !out || bad(B, N, M, S, C)) return VLP3D_EINVAL;
  const long long total = (long long)B * M * S * (C / 4 + 1);
  hipStream_t s = (hipStream_t)stream;
  if (out_bf16)
    hipLaunchKernelGGL((group_rows_fwd_kernel<__hip_bfloat16>), dim3(grid_for(total)), dim3(256), 0, s, xyz, new_xyz,
                       idx, feat_pm, N, M, S, C, radius, (__hip_bfloat16 *)out, total);
  else
    hipLaunchKernelGGL((group_rows_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, xyz, new_xyz, idx,
                       feat_pm, N, M, S, C, radius, (float *)out, total);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_group_rows_grad(const void *dout, int dout_bf16, const int *idx, int B, int N, int M, int S,
                                     int C, float radius, float *dfeat_pm, float *dxyz, float *dnew_xyz,
                                     void *stream) {
  if (!dout || !idx || bad(B, N, M, S, C)) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  if (dfeat_pm) e = vlp3d_zero_words(dfeat_pm, (size_t)B * N * C, s);
  if (e == hipSuccess && dxyz) e = vlp3d_zero_words(dxyz, (size_t)B * N * 3, s);
  if (e == hipSuccess && dnew_xyz) e = vlp3d_zero_words(dnew_xyz, (size_t)B * M * 3, s);
  if (e != hipSuccess) return (int)e;
  if (!dfeat_pm && !dxyz && !dnew_xyz) return VLP3D_OK;
  const long long total = (long long)B * M * S * (C / 4 + 1);
  if (dout_bf16)
    hipLaunchKernelGGL((group_rows_bwd_kernel<__hip_bfloat16>), dim3(grid_for(total)), dim3(256), 0, s,
                       (const __hip_bfloat16 *)dout, idx, N, M, S, C, radius, dfeat_pm, dxyz, dnew_xyz, total);
  else
    hipLaunchKernelGGL((group_rows_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float *)dout,
                       idx, N, M, S, C, radius, dfeat_pm, dxyz, dnew_xyz, total);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
