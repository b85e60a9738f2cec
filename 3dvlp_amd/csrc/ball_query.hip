// Ball query for gfx950 — replaces ball_query_gpu.cu:14-59 of the reference.
//
// Reference shape: one CUDA thread per centre, serial scan of all N points, 8 thread-blocks.
// Here: a wave owns C centres (wave-uniform, scalar registers); its 64 lanes test 64
// consecutive points per step (coalesced 768-byte reads of the L2-resident scene), a ballot
// turns hits into an ordered append (position = count + popcount(hits below my lane)) into an
// LDS-staged neighbour list, and the wave stops as soon as all of its centres are full.
// Rows are written back coalesced with the reference's padding rule (first hit repeated;
// all-zero row when the ball is empty).  Grid = B * ceil(M / (4*C)) workgroups of 4 waves,
// scene = blockIdx % B so that (for B a multiple or divisor of 8) the workgroups of one scene
// share one XCD's L2.
#include "common.h"

namespace {

template <int C>
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ new_xyz_all,
                                                         const float *__restrict__ xyz_all,
                                                         int *__restrict__ idx_all, int B, int N, int M,
                                                         float radius2, int nsample) {
  extern __shared__ int s_nb[];  // [4 waves][C][nsample]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // make it provably wave-uniform
  const int b = blockIdx.x % B;
  const int tile = blockIdx.x / B;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const float *__restrict__ new_xyz = new_xyz_all + (size_t)b * M * 3;
  int *__restrict__ nb = s_nb + (size_t)wave * C * nsample;
  const int j0 = (tile * 4 + wave) * C;

  float cx[C], cy[C], cz[C];
  int cnt[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = j0 + c;
    const bool ok = j < M;
    // wave-uniform loads (scalar)
    cx[c] = ok ? new_xyz[j * 3 + 0] : 0.f;
    cy[c] = ok ? new_xyz[j * 3 + 1] : 0.f;
    cz[c] = ok ? new_xyz[j * 3 + 2] : 0.f;
    cnt[c] = ok ? 0 : nsample;  // out-of-range centres are "full" from the start
  }

  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  float x = 0.f, y = 0.f, z = 0.f;
  if (lane < N) {
    x = xyz[lane * 3 + 0];
    y = xyz[lane * 3 + 1];
    z = xyz[lane * 3 + 2];
  }
  for (int k0 = 0; k0 < N; k0 += 64) {
    const int k = k0 + lane;
    const bool valid = k < N;
    // prefetch the next 64 points while this chunk is tested
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int kn = k + 64;
    if (kn < N) {
      nx = xyz[kn * 3 + 0];
      ny = xyz[kn * 3 + 1];
      nz = xyz[kn * 3 + 2];
    }
    bool all_full = true;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (cnt[c] < nsample) {  // wave-uniform
        const float d2 = vlp3d_sumsq3(cx[c] - x, cy[c] - y, cz[c] - z);
        const bool hit = valid && (d2 < radius2);
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
          const int pos = cnt[c] + __popcll(mask & lt_mask);
          if (hit && pos < nsample) nb[c * nsample + pos] = k;
          cnt[c] += __popcll(mask);
        }
        all_full = all_full && (cnt[c] >= nsample);
      }
    }
    if (all_full) break;
    x = nx; y = ny; z = nz;
  }
  __syncthreads();  // make this wave's LDS appends visible to its own later reads

#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = j0 + c;
    if (j < M) {  // wave-uniform
      const int n_hit = cnt[c] < nsample ? cnt[c] : nsample;
      const int first = n_hit > 0 ? nb[c * nsample] : 0;
      int *__restrict__ row = idx_all + ((size_t)b * M + j) * nsample;
      for (int l = lane; l < nsample; l += 64) row[l] = l < n_hit ? nb[c * nsample + l] : first;
    }
  }
}

}  // namespace

extern "C" int vlp3d_ball_query(const float *new_xyz, const float *xyz, int B, int N, int M, float radius,
                                int nsample, int *idx, void *stream) {
  if (!new_xyz || !xyz || !idx || B < 1 || N < 1 || M < 1 || nsample < 1) return VLP3D_EINVAL;
  if ((long long)N * 3 >= (1ll << 31) || (long long)M * 3 >= (1ll << 31)) return VLP3D_EINVAL;
  constexpr int C = 8;
  const size_t lds = (size_t)4 * C * nsample * sizeof(int);
  if (lds > 64 * 1024) return VLP3D_EINVAL;  // nsample <= 512
  const int tiles = vlp3d_cdiv(M, 4 * C);
  const long long grid = (long long)B * tiles;
  if (grid >= (1ll << 31)) return VLP3D_EINVAL;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27, fp32
  hipLaunchKernelGGL((ball_query_kernel<C>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, new_xyz,
                     xyz, idx, B, N, M, radius2, nsample);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
