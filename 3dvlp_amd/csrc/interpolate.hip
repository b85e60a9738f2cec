// three_nn / three_interpolate (+ adjoint) for gfx950 — replaces interpolate_gpu.cu:14-159.
#include "common.h"

namespace {

constexpr int KNOWN_TILE = 1024;  // known points staged per LDS pass (12 KiB)

// The known set streams through LDS in tiles shared by the workgroup.  The reference's strict `<` chain in
// index order makes the lowest index win ties (interpolate_gpu.cu:39-54): the order (distance, index).
// The reference keeps best* as double 1e40 and stores (float)best: +inf when fewer than 3 known
// points exist; float +inf reproduces both the comparisons and the stored value.
// Four lanes per unknown point (each walks every fourth known point in ascending order), branch-free insertion, then the
// four sorted triples are merged under the SAME total order the sequential strict-'<' chain produces: smaller distance first,
// equal distances by smaller index.  (One thread per unknown point with an if / else-if chain — the reference's shape —
// ran 61-76 us for 8 x 1024 x 512 on the geometry stream: 32 workgroups of divergent branches.)
__device__ __forceinline__ void nn3_insert(float d, int k, bool c1, bool c2, bool c3, float &b1, float &b2, float &b3, int &i1,
                                           int &i2, int &i3) {
  b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? k : i3);
  b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? k : i2);
  b1 = c1 ? d : b1;              i1 = c1 ? k : i1;
}

__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ unknown_all,
                                                       const float *__restrict__ known_all, int n, int m,
                                                       float *__restrict__ dist2_all, int *__restrict__ idx_all) {
  __shared__ float s_known[KNOWN_TILE * 3];
  const int b = blockIdx.y;
  const int part = threadIdx.x & 3;
  const int j = blockIdx.x * 64 + (threadIdx.x >> 2);
  const float *__restrict__ unknown = unknown_all + (size_t)b * n * 3;
  const float *__restrict__ known = known_all + (size_t)b * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) {
    ux = unknown[j * 3 + 0];
    uy = unknown[j * 3 + 1];
    uz = unknown[j * 3 + 2];
  }
  const float inf = __builtin_inff();
  float best1 = inf, best2 = inf, best3 = inf;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += KNOWN_TILE) {
    const int cnt = min(KNOWN_TILE, m - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += 256) s_known[t] = known[(size_t)k0 * 3 + t];
    __syncthreads();
    for (int kk = part; kk < cnt; kk += 4) {
      const float d = vlp3d_sumsq3(ux - s_known[kk * 3 + 0], uy - s_known[kk * 3 + 1], uz - s_known[kk * 3 + 2]);
      nn3_insert(d, k0 + kk, d < best1, d < best2, d < best3, best1, best2, best3, besti1, besti2, besti3);
    }
  }
  // merge: every lane inserts the triples of the other three lanes of its quad (xor 1, then xor 2 of the merged result)
#pragma unroll
  for (int step = 1; step <= 2; step <<= 1) {
    const float o1 = __shfl_xor(best1, step), o2 = __shfl_xor(best2, step), o3 = __shfl_xor(best3, step);
    const int p1 = __shfl_xor(besti1, step), p2 = __shfl_xor(besti2, step), p3 = __shfl_xor(besti3, step);
    const float od[3] = {o1, o2, o3};
    const int ok[3] = {p1, p2, p3};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const float d = od[e];
      const int k = ok[e];
      // an unfilled slot is (+inf, 0): it never displaces anything (inf < x is false; inf == inf needs k < 0)
      const bool c1 = d < best1 || (d == best1 && k < besti1);
      const bool c2 = d < best2 || (d == best2 && k < besti2);
      const bool c3 = d < best3 || (d == best3 && k < besti3);
      nn3_insert(d, k, c1, c2, c3, best1, best2, best3, besti1, besti2, besti3);
    }
  }
  if (j < n && part == 0) {
    float *__restrict__ d2 = dist2_all + ((size_t)b * n + j) * 3;
    int *__restrict__ id = idx_all + ((size_t)b * n + j) * 3;
    d2[0] = best1; d2[1] = best2; d2[2] = best3;
    id[0] = besti1; id[1] = besti2; id[2] = besti3;
  }
}

// out[b,c,j] = p[c,i1]*w1 + p[c,i2]*w2 + p[c,i3]*w3 ; thread owns j, walks a slab of channels.
__global__ __launch_bounds__(256) void three_interpolate_kernel(const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight, int C, int m,
                                                                int n, int c_per_block, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *__restrict__ id = idx + ((size_t)b * n + j) * 3;
  const float *__restrict__ w = weight + ((size_t)b * n + j) * 3;
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) {
    const float *__restrict__ p = points + ((size_t)b * C + c) * m;
    out[((size_t)b * C + c) * n + j] = vlp3d_blend3(p[i1], w1, p[i2], w2, p[i3], w3);
  }
}

// True adjoint (what interpolate_gpu.cu:121-148 intends): grad_points[b,c,i_t] += grad_out[b,c,j]*w_t.
__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight, int C, int n,
                                                                     int m, int c_per_block,
                                                                     float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *__restrict__ id = idx + ((size_t)b * n + j) * 3;
  const float *__restrict__ w = weight + ((size_t)b * n + j) * 3;
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) {
    const float g = grad_out[((size_t)b * C + c) * n + j];
    float *__restrict__ gp = grad_points + ((size_t)b * C + c) * m;
    atomicAdd(gp + i1, g * w1);
    atomicAdd(gp + i2, g * w2);
    atomicAdd(gp + i3, g * w3);
  }
}

// LDS-privatised form for m <= 2048 known points: a workgroup owns CPB channels of one scene, accumulates their m
// gradients in LDS (ds_add_f32; the global-atomic form runs ~6 colliding adds per address through L2: 103 us at
// (8,256,1024)->(8,256,512)) and writes every output exactly once — no memset, no global atomics.
template <int CPB>
__global__ __launch_bounds__(256) void three_interpolate_grad_lds_kernel(const float *__restrict__ grad_out,
                                                                         const int *__restrict__ idx,
                                                                         const float *__restrict__ weight, int C, int n,
                                                                         int m, float *__restrict__ grad_points) {
  extern __shared__ float acc[];  // [CPB][m]
  const int b = blockIdx.y, c0 = blockIdx.x * CPB;
  for (int i = threadIdx.x; i < CPB * m; i += 256) acc[i] = 0.f;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256) {
    const int *__restrict__ id = idx + ((size_t)b * n + j) * 3;
    const float *__restrict__ w = weight + ((size_t)b * n + j) * 3;
    const int i1 = id[0], i2 = id[1], i3 = id[2];
    const float w1 = w[0], w2 = w[1], w3 = w[2];
#pragma unroll
    for (int cc = 0; cc < CPB; ++cc) {
      if (c0 + cc >= C) break;
      const float g = grad_out[((size_t)b * C + c0 + cc) * n + j];
      atomicAdd(acc + cc * m + i1, g * w1);
      atomicAdd(acc + cc * m + i2, g * w2);
      atomicAdd(acc + cc * m + i3, g * w3);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CPB * m; i += 256) {
    const int cc = i / m;
    if (c0 + cc < C) grad_points[((size_t)b * C + c0 + cc) * m + (i - cc * m)] = acc[i];
  }
}

int pick_c_per_block(int C, long long blocks_xz) {
  int split = (int)((2048 + blocks_xz - 1) / blocks_xz);
  if (split < 1) split = 1;
  if (split > C) split = C;
  return (C + split - 1) / split;
}

}  // namespace

extern "C" int vlp3d_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2, int *idx,
                              void *stream) {
  if (!unknown || !known || !dist2 || !idx || B < 1 || B > 65535 || n < 1 || m < 1) return VLP3D_EINVAL;
  hipLaunchKernelGGL(three_nn_kernel, dim3(vlp3d_cdiv(n, 64), B), dim3(256), 0, (hipStream_t)stream, unknown, known,
                     n, m, dist2, idx);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_three_interpolate(const float *points, const int *idx, const float *weight, int B, int C, int m,
                                       int n, float *out, void *stream) {
  if (!points || !idx || !weight || !out || B < 1 || B > 65535 || C < 1 || m < 1 || n < 1) return VLP3D_EINVAL;
  const int gx = vlp3d_cdiv(n, 256);
  const int cpb = pick_c_per_block(C, (long long)gx * B);
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(gx, vlp3d_cdiv(C, cpb), B), dim3(256), 0, (hipStream_t)stream,
                     points, idx, weight, C, m, n, cpb, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_three_interpolate_grad(const float *grad_out, const int *idx, const float *weight, int B, int C,
                                            int n, int m, float *grad_points, void *stream) {
  if (!grad_out || !idx || !weight || !grad_points || B < 1 || B > 65535 || C < 1 || m < 1 || n < 1)
    return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (m <= 2048) {
    constexpr int CPB = 4;
    hipLaunchKernelGGL(three_interpolate_grad_lds_kernel<CPB>, dim3(vlp3d_cdiv(C, CPB), B), dim3(256),
                       sizeof(float) * CPB * m, s, grad_out, idx, weight, C, n, m, grad_points);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  hipError_t e = vlp3d_zero_words(grad_points, (size_t)B * C * m, s);
  if (e != hipSuccess) return (int)e;
  const int gx = vlp3d_cdiv(n, 256);
  const int cpb = pick_c_per_block(C, (long long)gx * B);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(gx, vlp3d_cdiv(C, cpb), B), dim3(256), 0, s, grad_out, idx,
                     weight, C, n, m, cpb, grad_points);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}


// ---- loader-side glue of the geometry stream (no gradient, coordinates only) ----------------------------------------------
// gather_xyz: new_xyz[b,j,:] = xyz[b, idx[b,j], :] on the (B,N,3) layout the geometry kernels use — replaces
//   gather_operation(xyz.transpose(1,2).contiguous(), idx).transpose(1,2).contiguous() (two 480 KB transposes + a gather per
//   backbone level; pointnet2_modules.py:233-236 does exactly that).
// three_nn_weights: the inverse-distance weights of PointnetFPModule.forward (pointnet2_modules.py:393-397) from three_nn's
//   squared distances in one launch: dist = sqrt(d2); r = 1/(dist + 1e-8); w = r / (r0 + r1 + r2)  (five element-wise
//   launches in the reference's op sequence); `dist` (optional) receives sqrt(d2) like ThreeNN.forward returns it.
namespace {
__global__ __launch_bounds__(256) void gather_xyz_kernel(const float *__restrict__ xyz, const int *__restrict__ idx, int N, int M,
                                                         long long total, float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long b = i / M;
  const float *p = xyz + (b * N + idx[i]) * 3;
  out[i * 3] = p[0]; out[i * 3 + 1] = p[1]; out[i * 3 + 2] = p[2];
}
__global__ __launch_bounds__(256) void three_nn_weights_kernel(const float *__restrict__ d2, long long n, float *__restrict__ w,
                                                               float *__restrict__ dist) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = sqrtf(d2[3 * i]), b = sqrtf(d2[3 * i + 1]), c = sqrtf(d2[3 * i + 2]);
  const float ra = 1.0f / (a + 1e-8f), rb = 1.0f / (b + 1e-8f), rc = 1.0f / (c + 1e-8f);
  const float norm = (ra + rb) + rc;
  w[3 * i] = ra / norm; w[3 * i + 1] = rb / norm; w[3 * i + 2] = rc / norm;
  if (dist) { dist[3 * i] = a; dist[3 * i + 1] = b; dist[3 * i + 2] = c; }
}
}  // namespace

extern "C" int vlp3d_gather_xyz(const float *xyz, const int *idx, int B, int N, int M, float *out, void *stream) {
  if (!xyz || !idx || !out || B < 1 || N < 1 || M < 1) return -22;
  const long long total = (long long)B * M;
  hipLaunchKernelGGL(gather_xyz_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, idx, N, M,
                     total, out);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void gather_xyz_grad_kernel(const float *__restrict__ g, const int *__restrict__ idx, int N, int M,
                                                              long long total, float *__restrict__ dxyz) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float *p = dxyz + ((i / M) * N + idx[i]) * 3;
  unsafeAtomicAdd(p, g[i * 3]); unsafeAtomicAdd(p + 1, g[i * 3 + 1]); unsafeAtomicAdd(p + 2, g[i * 3 + 2]);
}
}  // namespace

// adjoint of vlp3d_gather_xyz: dxyz (B,N,3) = 0, then dxyz[b, idx[b,j], :] += g[b,j,:] (duplicated indices accumulate)
extern "C" int vlp3d_gather_xyz_grad(const float *g, const int *idx, int B, int N, int M, float *dxyz, void *stream) {
  if (!g || !idx || !dxyz || B < 1 || N < 1 || M < 1) return -22;
  hipError_t e = vlp3d_zero_words(dxyz, (size_t)B * N * 3, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  const long long total = (long long)B * M;
  hipLaunchKernelGGL(gather_xyz_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, idx, N, M,
                     total, dxyz);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlp3d_three_nn_weights(const float *dist2, long long n, float *weight, float *dist, void *stream) {
  if (!dist2 || !weight || n < 1) return -22;
  hipLaunchKernelGGL(three_nn_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dist2, n,
                     weight, dist);
  VLP3D_LAUNCH_CHECK();
  return 0;
}
