// Furthest point sampling for gfx950 — replaces sampling_gpu.cu:74-234 of the reference.
//
// Design: ONE workgroup per scene (the m-1 selections are a serial chain),
// but unlike the reference the point set does not stream from memory every iteration:
// each thread owns the points k = tid + i*THREADS and keeps (x, y, z, running-min) of its
// first R of them in VGPRs, the next L in LDS (one float4 per point, ds_read_b128) and only
// the remainder in global memory (L2-resident).  The per-iteration block argmax is a
// single 64-bit key max: DPP inside the wave, one ds_max_u64 per wave, one barrier.
//
// Bit-exact selection order: the reference's 512-slot LDS tree (sampling_gpu.cu:64-70,116-175)
// resolves equal maxima towards the smallest (bitrev_P(k mod P), k), P = opt_n_threads(n).
// The key below encodes (value, that order) so that any ownership/reduction shape gives the
// same winner:  key = (float_bits(best) + 1) << 32 | ~(brev32(k & (P-1)) | (k >> log2 P)).
// key == 0 encodes "no candidate" (best stayed -1), which the reference resolves to index 0.
#include "common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {
  unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
  unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o > v ? o : v;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// max over the 64 lanes of a wave; result valid (and wave-uniform) in every lane.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_u64<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_max_u64<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_max_u64<0x141>(v);  // row_half_mirror
  v = dpp_max_u64<0x140>(v);  // row_mirror  -> each row of 16 lanes holds its max
  unsigned long long a = readlane_u64(v, 0), b = readlane_u64(v, 16);
  unsigned long long c = readlane_u64(v, 32), d = readlane_u64(v, 48);
  a = a > b ? a : b;
  c = c > d ? c : d;
  return a > c ? a : c;
}

// fminf without the canonicalising v_max hipcc puts in front of it (inputs here are never sNaN;
// like CUDA's min(), v_min_f32 returns the non-NaN operand).
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int THREADS, int R, int L>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float *__restrict__ xyz_all,
                                                      float *__restrict__ temp_all,
                                                      int *__restrict__ idx_all, int N, int m, int log2P,
                                                      const int *__restrict__ not_prefix) {
  __shared__ unsigned long long s_best[3];
  __shared__ float4 s_pts[L > 0 ? L * THREADS : 1];
  if (not_prefix != nullptr && *not_prefix == 0) {
    // fps_prefix_check proved that the sampling order is 0, 1, 2, ..: the points are the output of an earlier FPS, in
    // its order, and no step has a tie (see below)
    int *__restrict__ out = idx_all + (size_t)blockIdx.x * m;
    for (int j = threadIdx.x; j < m; j += THREADS) out[j] = j;
    return;
  }

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  float *__restrict__ temp = temp_all + (size_t)b * N;
  int *__restrict__ idx = idx_all + (size_t)b * m;
  const int PPT = (N + THREADS - 1) / THREADS;

  float px[R], py[R], pz[R], pt[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int k = tid + i * THREADS;
    float x = 0.f, y = 0.f, z = 0.f, t = -1.f;  // t = -1: never a candidate (d2 > best fails)
    if (k < N) {
      x = xyz[k * 3 + 0];
      y = xyz[k * 3 + 1];
      z = xyz[k * 3 + 2];
      t = vlp3d_fps_skipped(x, y, z) ? -1.f : 1e10f;
    }
    px[i] = x; py[i] = y; pz[i] = z; pt[i] = t;
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int k = tid + (R + i) * THREADS;
    float x = 0.f, y = 0.f, z = 0.f, t = -1.f;
    if (k < N) {
      x = xyz[k * 3 + 0];
      y = xyz[k * 3 + 1];
      z = xyz[k * 3 + 2];
      t = vlp3d_fps_skipped(x, y, z) ? -1.f : 1e10f;
    }
    s_pts[i * THREADS + tid] = make_float4(x, y, z, t);
  }
  for (int i = R + L; i < PPT; ++i) {
    const int k = tid + i * THREADS;
    if (k < N) temp[k] = vlp3d_fps_skipped(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]) ? -1.f : 1e10f;
  }
  if (tid < 3) s_best[tid] = 0ull;
  if (tid == 0) idx[0] = 0;
  __syncthreads();

  const unsigned Pm1 = (1u << log2P) - 1u;
  const unsigned lowmask = (unsigned)((1ull << (32 - log2P)) - 1ull);
  int old = 0;
  int slot = 1;
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0];
    const float y1 = xyz[old * 3 + 1];
    const float z1 = xyz[old * 3 + 2];
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float d = vlp3d_sumsq3(px[i] - x1, py[i] - y1, pz[i] - z1);
      const float t = vmin(d, pt[i]);
      pt[i] = t;
      const bool g = t > best;
      bi = g ? i : bi;
      best = g ? t : best;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const float4 p = s_pts[i * THREADS + tid];
      const float d = vlp3d_sumsq3(p.x - x1, p.y - y1, p.z - z1);
      const float t = vmin(d, p.w);
      s_pts[i * THREADS + tid].w = t;
      const bool g = t > best;
      bi = g ? (R + i) : bi;
      best = g ? t : best;
    }
    for (int i = R + L; i < PPT; ++i) {
      const int k = tid + i * THREADS;
      if (k < N) {
        const float d = vlp3d_sumsq3(xyz[k * 3 + 0] - x1, xyz[k * 3 + 1] - y1, xyz[k * 3 + 2] - z1);
        const float t = vmin(d, temp[k]);
        temp[k] = t;
        const bool g = t > best;
        bi = g ? i : bi;
        best = g ? t : best;
      }
    }

    unsigned long long key = 0ull;
    if (best >= 0.f) {
      const unsigned k = (unsigned)tid + (unsigned)bi * THREADS;
      const unsigned tie = __brev(k & Pm1) | (k >> log2P);
      key = ((unsigned long long)(__float_as_uint(best) + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - tie);
    }
    key = wave_max_u64(key);
    if ((tid & 63) == 0) atomicMax(&s_best[slot], key);
    const int nslot = slot == 2 ? 0 : slot + 1;
    if (tid == 0) s_best[nslot] = 0ull;  // last read two barriers ago; next written after this barrier
    __syncthreads();
    const unsigned long long g = s_best[slot];
    slot = nslot;

    unsigned k = 0u;
    if ((unsigned)(g >> 32) != 0u) {
      const unsigned tie = 0xFFFFFFFFu - (unsigned)g;
      k = ((tie & lowmask) << log2P) | __brev(tie & ~lowmask);
    }
    old = __builtin_amdgcn_readfirstlane((int)k);
    if (tid == 0) idx[j] = old;
  }
}

// include/cuda_utils.h:20-24 of the reference: block size used by its FPS launch.
int reference_log2_block(int n) {
  int p = (int)(log((double)n) / log(2.0));
  if (p > 9) p = 9;
  if (p < 0) p = 0;
  return p;
}

}  // namespace

static int fps_dense(const float *xyz, int B, int N, int m, float *temp, int *idx, const int *not_prefix, void *stream) {
  if (!xyz || !temp || !idx || B < 1 || N < 1 || m < 0) return VLP3D_EINVAL;
  if ((long long)N * 3 >= (1ll << 31)) return VLP3D_EINVAL;
  if (m == 0) return VLP3D_OK;
  hipStream_t s = (hipStream_t)stream;
  const int log2P = reference_log2_block(N);
#define FPS_LAUNCH(T, R, L) \
  hipLaunchKernelGGL((fps_kernel<T, R, L>), dim3(B), dim3(T), 0, s, xyz, temp, idx, N, m, log2P, not_prefix)
  // THREADS must be a multiple of P = 2^log2P (the reference's block size) so that all points of
  // one thread share k mod P: then "lowest slot wins" inside a thread is the reference's order.
  if (N < 512) FPS_LAUNCH(256, 2, 0);        // P <= 256
  else if (N <= 1024) FPS_LAUNCH(512, 2, 0);  // P == 512 from here on (or 256 at N == 512)
  else if (N <= 2048) FPS_LAUNCH(512, 4, 0);
  else if (N <= 4096) FPS_LAUNCH(1024, 4, 0);
  else if (N <= 8192) FPS_LAUNCH(1024, 8, 0);
  else if (N <= 16384) FPS_LAUNCH(1024, 16, 0);
  else FPS_LAUNCH(1024, 24, 9);
#undef FPS_LAUNCH
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_furthest_point_sampling(const float *xyz, int B, int N, int m, float *temp, int *idx,
                                             void *stream) {
  return fps_dense(xyz, B, N, m, temp, idx, nullptr, stream);
}

// ---- FPS of an FPS-ordered point set ---------------------------------------------------------------------------------
// The backbone samples every level from the PREVIOUS level's samples, which are stored in sampling order
// (backbone_module.py:93-117; its fp2_inds = sa1_inds[:, :num_seed] relies on the same fact): greedy FPS restricted to a
// prefix-closed subset that starts at the same point picks 0, 1, 2, ... again — the j-th sample maximises the distance to
// samples 0..j-1 over ALL points, hence over the subset.  That makes m-1 dependent block-wide argmaxes (0.6 + 0.2 + 0.2 ms
// for SA2..SA4 at cfg2, one workgroup per scene) replaceable by a fully parallel PROOF: for every step j < m and every
// point p > j,   min_{q<j} d(p, q)  <  v_j = min_{q<j} d(j, q)   strictly, v_j > 0, and no point in the skip ball.
// Strict inequalities leave no tie for the reference's reduction order to decide, and the distances are the sequential
// kernel's own expression (vlp3d_sumsq3(p - q)), bit for bit.  If anything fails (not FPS-ordered input, a tie, a skipped
// point) *not_prefix becomes 1 and the conditional entry below runs the sequential kernel: exact either way.
namespace {
// Both kernels walk the samples in chunks of 256 staged in LDS (x, y, z, and for the check the bound v of the NEXT step):
// one broadcast ds_read_b128 per sample instead of a dependent global load per loop iteration (0.38 + 0.19 ms -> see
// DESIGN §4.1 for the three levels of cfg2).
__global__ __launch_bounds__(256) void fps_prefix_v_kernel(const float *__restrict__ xyz_all, int N, int m,
                                                           float *__restrict__ v_all, int *__restrict__ not_prefix) {
  __shared__ float4 sm[256];
  const int b = blockIdx.y, i0 = blockIdx.x * 256, i = i0 + threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *not_prefix = 0;  // the check kernel runs after this one
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const int ic = min(i, m - 1);
  const float x = xyz[ic * 3], y = xyz[ic * 3 + 1], z = xyz[ic * 3 + 2];
  float v = 1e10f;
  const int qend = min(i0 + 255, m - 1);  // samples q < i for the largest i of the block
  for (int c0 = 0; c0 < qend; c0 += 256) {
    const int q = min(c0 + (int)threadIdx.x, m - 1);
    __syncthreads();
    sm[threadIdx.x] = make_float4(xyz[q * 3], xyz[q * 3 + 1], xyz[q * 3 + 2], 0.f);
    __syncthreads();
    const int n = min(256, qend - c0);
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      const float4 s = sm[k];
      const float d = vlp3d_sumsq3(x - s.x, y - s.y, z - s.z);
      v = (c0 + k < i) ? fminf(v, d) : v;
    }
  }
  if (i < m) v_all[(size_t)b * m + i] = v;
}

__global__ __launch_bounds__(256) void fps_prefix_check_kernel(const float *__restrict__ xyz_all, int N, int m,
                                                               const float *__restrict__ v_all,
                                                               int *__restrict__ not_prefix) {
  __shared__ float4 sm[256];
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const float *__restrict__ v = v_all + (size_t)b * m;
  const int pc = min(p, N - 1);
  const float x = xyz[pc * 3], y = xyz[pc * 3 + 1], z = xyz[pc * 3 + 2];
  bool bad = vlp3d_fps_skipped(x, y, z);
  if (p >= 1 && p < m) bad = bad || !(v[p] > 0.f);
  // step j (1 <= j < m) happens while p is unselected iff j < steps; after sample q = j - 1 the running minimum r is
  // min over samples <= q, to be compared with v[q + 1]
  const int steps = p < m ? p : m;
  float r = 1e10f;
  for (int c0 = 0; c0 < m - 1; c0 += 256) {
    const int q = min(c0 + (int)threadIdx.x, m - 2);
    __syncthreads();
    sm[threadIdx.x] = make_float4(xyz[q * 3], xyz[q * 3 + 1], xyz[q * 3 + 2], v[q + 1]);
    __syncthreads();
    const int n = min(256, m - 1 - c0);
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      const float4 s = sm[k];
      r = fminf(r, vlp3d_sumsq3(x - s.x, y - s.y, z - s.z));
      bad = bad || (c0 + k + 1 < steps && !(r < s.w));
    }
  }
  if (bad && p < N) *not_prefix = 1;
}
}  // namespace

// v: (B, m) floats of scratch; not_prefix: one int (0 = "the sampling order is 0..m-1" proven).  N, m <= 65536.
extern "C" int vlp3d_fps_prefix_check(const float *xyz, int B, int N, int m, float *v, int *not_prefix, void *stream) {
  if (!xyz || !v || !not_prefix || B < 1 || N < 1 || m < 1 || m > N || N > 65536) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fps_prefix_v_kernel, dim3((m + 255) / 256, B), dim3(256), 0, s, xyz, N, m, v, not_prefix);
  hipLaunchKernelGGL(fps_prefix_check_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, xyz, N, m, v, not_prefix);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// vlp3d_furthest_point_sampling, skipped (idx = 0..m-1) when *not_prefix == 0 (vlp3d_fps_prefix_check ran before on the
// same stream); not_prefix == NULL: unconditional.
extern "C" int vlp3d_furthest_point_sampling_cond(const float *xyz, int B, int N, int m, float *temp, int *idx,
                                                  const int *not_prefix, void *stream) {
  return fps_dense(xyz, B, N, m, temp, idx, not_prefix, stream);
}
