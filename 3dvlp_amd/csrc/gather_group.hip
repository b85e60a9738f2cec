// gather_points / group_points (+ their scatter-add adjoints) for gfx950 — replaces
// sampling_gpu.cu:13-62 and group_points_gpu.cu:13-80 of the reference.
//
// The reference launches one block per scene (8 blocks on the whole GPU).  Here the grid is
// flat over output elements: a thread owns one output position (b, j[,k]) — its index is read
// once — and walks the C channels, so writes are coalesced along the fastest output dimension
// and the launch fills all 256 CUs.
#include "common.h"

namespace {

// out[b,c,j] = points[b,c,idx[b,j]]
__global__ __launch_bounds__(256) void gather_points_kernel(const float *__restrict__ points,
                                                            const int *__restrict__ idx, int C, int N, int M,
                                                            int c_per_block, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const int a = idx[(size_t)b * M + j];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) out[((size_t)b * C + c) * M + j] = points[((size_t)b * C + c) * N + a];
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(const float *__restrict__ grad_out,
                                                                 const int *__restrict__ idx, int C, int N, int M,
                                                                 int c_per_block, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const int a = idx[(size_t)b * M + j];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c)
    atomicAdd(grad_points + ((size_t)b * C + c) * N + a, grad_out[((size_t)b * C + c) * M + j]);
}

// out[b,c,j,k] = points[b,c,idx[b,j,k]];  e = j*S + k is the flat position inside a (M,S) plane.
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ points,
                                                           const int *__restrict__ idx, int C, int N, int MS,
                                                           int c_per_block, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= MS) return;
  const int a = idx[(size_t)b * MS + e];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  const float *__restrict__ src = points + (size_t)b * C * N + a;
  float *__restrict__ dst = out + (size_t)b * C * MS + e;
#pragma unroll 4
  for (int c = c0; c < c1; ++c) dst[(size_t)c * MS] = src[(size_t)c * N];
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(const float *__restrict__ grad_out,
                                                                const int *__restrict__ idx, int C, int N, int MS,
                                                                int c_per_block, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= MS) return;
  const int a = idx[(size_t)b * MS + e];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  const float *__restrict__ src = grad_out + (size_t)b * C * MS + e;
  float *__restrict__ dst = grad_points + (size_t)b * C * N + a;
#pragma unroll 4
  for (int c = c0; c < c1; ++c) atomicAdd(dst + (size_t)c * N, src[(size_t)c * MS]);
}

// Channels per block: enough blocks to fill the chip (>= ~2048) without starving each of work.
int pick_c_per_block(int C, long long blocks_xz) {
  int split = (int)((2048 + blocks_xz - 1) / blocks_xz);
  if (split < 1) split = 1;
  if (split > C) split = C;
  return (C + split - 1) / split;
}

bool bad_dims(int B, int C, int N, int M) { return B < 1 || C < 1 || N < 1 || M < 1 || B > 65535; }

}  // namespace

extern "C" int vlp3d_gather_points(const float *points, const int *idx, int B, int C, int N, int M, float *out,
                                   void *stream) {
  if (!points || !idx || !out || bad_dims(B, C, N, M)) return VLP3D_EINVAL;
  const int gx = vlp3d_cdiv(M, 256);
  const int cpb = pick_c_per_block(C, (long long)gx * B);
  hipLaunchKernelGGL(gather_points_kernel, dim3(gx, vlp3d_cdiv(C, cpb), B), dim3(256), 0, (hipStream_t)stream, points,
                     idx, C, N, M, cpb, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_gather_points_grad(const float *grad_out, const int *idx, int B, int C, int N, int M,
                                        float *grad_points, void *stream) {
  if (!grad_out || !idx || !grad_points || bad_dims(B, C, N, M)) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = vlp3d_zero_words(grad_points, (size_t)B * C * N, s);
  if (e != hipSuccess) return (int)e;
  const int gx = vlp3d_cdiv(M, 256);
  const int cpb = pick_c_per_block(C, (long long)gx * B);
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(gx, vlp3d_cdiv(C, cpb), B), dim3(256), 0, s, grad_out, idx, C, N,
                     M, cpb, grad_points);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_group_points(const float *points, const int *idx, int B, int C, int N, int M, int S,
                                  float *out, void *stream) {
  if (!points || !idx || !out || bad_dims(B, C, N, M) || S < 1 || (long long)M * S >= (1ll << 31)) return VLP3D_EINVAL;
  const int MS = M * S;
  const int gx = vlp3d_cdiv(MS, 256);
  const int cpb = pick_c_per_block(C, (long long)gx * B);
  hipLaunchKernelGGL(group_points_kernel, dim3(gx, vlp3d_cdiv(C, cpb), B), dim3(256), 0, (hipStream_t)stream, points,
                     idx, C, N, MS, cpb, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_group_points_grad(const float *grad_out, const int *idx, int B, int C, int N, int M, int S,
                                       float *grad_points, void *stream) {
  if (!grad_out || !idx || !grad_points || bad_dims(B, C, N, M) || S < 1 || (long long)M * S >= (1ll << 31))
    return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = vlp3d_zero_words(grad_points, (size_t)B * C * N, s);
  if (e != hipSuccess) return (int)e;
  const int MS = M * S;
  const int gx = vlp3d_cdiv(MS, 256);
  const int cpb = pick_c_per_block(C, (long long)gx * B);
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(gx, vlp3d_cdiv(C, cpb), B), dim3(256), 0, s, grad_out, idx, C, N,
                     MS, cpb, grad_points);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
