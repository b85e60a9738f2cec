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
    // LDS-only barrier (the global stores of this loop — temp of the overflow points, idx — are read by nobody but their
    // own thread before the kernel ends)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// ---- small point sets (N <= 2048: the proposal sampling of the vote clusters, 1024 -> 256 on the step's critical path) ----
// The iteration is nothing but its dependency chain here (update of <= 8 points per lane, then argmax), so the chain is
// cut to the pruned kernel's form: four waves only; 32-bit value reductions (DPP) with the reference's tie order resolved
// lazily — only when a maximum is attained twice do the tied lanes compare the tie key; the wave candidates (value, index)
// meet in LDS behind an LDS-only barrier and every wave reduces the four of them redundantly (no second barrier, no LDS
// atomics); coordinates and the sample list live in LDS.  256 threads are HALF the reference's block P = 512 at
// 512 <= N <= 2048, so a lane's points k = tid + 256 i alternate between two residues mod P: walking the even slots first,
// then the odd ones, with a strict '>' keeps "first maximum wins" equal to the reference's order
// (smallest (bitrev_P(k mod P), k)); below 512 points 256 is a multiple of P and the natural order is that order.
// 0.54 -> 0.3 us per iteration at 8 x 1024 -> 256.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32s(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32s(unsigned v) {  // wave-uniform result
  v = dpp_max_u32s<0xB1>(v);
  v = dpp_max_u32s<0x4E>(v);
  v = dpp_max_u32s<0x141>(v);
  v = dpp_max_u32s<0x140>(v);
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

template <int PPT>
__global__ __launch_bounds__(256) void fps_small_kernel(const float *__restrict__ xyz_all, int *__restrict__ idx_all, int N,
                                                        int m, int log2P, const int *__restrict__ not_prefix) {
  __shared__ uint2 s_vk[2][4];
  __shared__ float4 s_xyz[PPT * 256];
  __shared__ int s_idx[PPT * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int *__restrict__ idx = idx_all + (size_t)blockIdx.x * m;
  if (not_prefix != nullptr && *not_prefix == 0) {  // see fps_kernel
    for (int j = tid; j < m; j += 256) idx[j] = j;
    return;
  }
  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  const bool evenodd = log2P == 9;  // P = 512 = 2 x 256 threads
  const unsigned Pm1 = (1u << log2P) - 1u;
  auto slot_of = [&](int u) -> int { return evenodd ? (u < PPT / 2 ? 2 * u : 2 * (u - PPT / 2) + 1) : u; };
  auto tiekey = [&](unsigned k) -> unsigned { return 0xFFFFFFFFu - (__brev(k & Pm1) | (k >> log2P)); };  // larger = preferred

  float px[PPT], py[PPT], pz[PPT], pt[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int k = tid + 256 * slot_of(u);
    float x = 0.f, y = 0.f, z = 0.f, t = -1.f;  // t = -1: never a candidate
    if (k < N) {
      x = xyz[k * 3 + 0]; y = xyz[k * 3 + 1]; z = xyz[k * 3 + 2];
      t = vlp3d_fps_skipped(x, y, z) ? -1.f : 1e10f;
      s_xyz[k] = make_float4(x, y, z, 0.f);
    }
    px[u] = x; py[u] = y; pz[u] = z; pt[u] = t;
  }
  __syncthreads();
  int old = 0, par = 0;
  for (int j = 1; j < m; ++j) {
    const float4 q = s_xyz[old];
    float best = -1.f;
    int bu = 0;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const float d = vlp3d_sumsq3(px[u] - q.x, py[u] - q.y, pz[u] - q.z);
      const float t = vmin(d, pt[u]);
      pt[u] = t;
      const bool g = t > best;
      bu = g ? u : bu;
      best = g ? t : best;
    }
    const unsigned k = (unsigned)(tid + 256 * slot_of(bu));
    const unsigned v = best >= 0.f ? __float_as_uint(best) + 1u : 0u;
    const unsigned vmax = wave_max_u32s(v);
    unsigned long long tied = __ballot(v == vmax);
    if (__popcll(tied) > 1) {  // wave-uniform, rare: an exact tie — the reference's order decides
      const unsigned tk = v == vmax ? tiekey(k) : 0u;
      const unsigned tmax = wave_max_u32s(tk);
      tied = __ballot(v == vmax && tk == tmax);
    }
    const unsigned kw = (unsigned)__builtin_amdgcn_readlane((int)k, (int)__builtin_ctzll(tied));
    if (lane == 0) s_vk[par][wave] = make_uint2(vmax, kw);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only barrier (see fps_kernel)
    const uint2 c = s_vk[par][lane & 3];
    const unsigned bv = lane < 4 ? c.x : 0u;
    unsigned r = dpp_max_u32s<0xB1>(bv);
    r = dpp_max_u32s<0x4E>(r);
    const unsigned bmax = (unsigned)__builtin_amdgcn_readlane((int)r, 0);
    unsigned knew = 0u;  // no candidate left (every point skipped): the reference returns index 0
    if (bmax != 0u) {
      unsigned long long w4 = __ballot(lane < 4 && bv == bmax);
      if (__popcll(w4) > 1) {
        const unsigned tk = (lane < 4 && bv == bmax) ? tiekey(c.y) : 0u;
        unsigned tm = dpp_max_u32s<0xB1>(tk);
        tm = dpp_max_u32s<0x4E>(tm);
        const unsigned tmax = (unsigned)__builtin_amdgcn_readlane((int)tm, 0);
        w4 = __ballot(lane < 4 && bv == bmax && tk == tmax);
      }
      knew = (unsigned)__builtin_amdgcn_readlane((int)c.y, (int)__builtin_ctzll(w4));
    }
    old = (int)knew;
    if (tid == 0) s_idx[j] = old;
    par ^= 1;
  }
  __syncthreads();
  if (tid == 0) idx[0] = 0;
  for (int j = 1 + tid; j < m; j += 256) idx[j] = s_idx[j];
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
  // fps_kernel: THREADS must be a multiple of P = 2^log2P (the reference's block size) so that all points of
  // one thread share k mod P: then "lowest slot wins" inside a thread is the reference's order.
#define FPS_LAUNCH_S(PPT) \
  hipLaunchKernelGGL((fps_small_kernel<PPT>), dim3(B), dim3(256), 0, s, xyz, idx, N, m, log2P, not_prefix)
  if (N <= 256) FPS_LAUNCH_S(1);
  else if (N <= 512) FPS_LAUNCH_S(2);
  else if (N <= 1024) FPS_LAUNCH_S(4);
  else if (N <= 2048) FPS_LAUNCH_S(8);
  else if (N <= 4096) FPS_LAUNCH(1024, 4, 0);
  else if (N <= 8192) FPS_LAUNCH(1024, 8, 0);
  else if (N <= 16384) FPS_LAUNCH(1024, 16, 0);
  else FPS_LAUNCH(1024, 24, 9);
#undef FPS_LAUNCH
#undef FPS_LAUNCH_S
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

// The same two passes for m <= 2048 samples with EIGHT lanes per point / sample and every sample of the scene in LDS at once
// (the forms above give a lane 1024 dependent iterations and launch 32-64 workgroups: 62 + 78 us for the three levels of cfg2
// on the geometry stream, which is the step's critical path since the split backward).  v: the lanes of a group take every
// eighth earlier sample.  check: the running minimum is a prefix minimum over the samples, so a group splits the samples that
// matter for its point (q < steps - 1) into eight contiguous parts: each lane first takes the minimum of its part, an exclusive
// prefix minimum over the group gives its starting value, then it walks its part again with the comparisons.
constexpr int PFX_LANES = 8;  // 16 / 32 lanes: 57 -> 68 us at 2048 -> 1024 (every workgroup stages all samples), 14 -> 9 us at 512 -> 256
__global__ __launch_bounds__(256) void fps_prefix_v8_kernel(const float *__restrict__ xyz_all, int N, int m,
                                                            float *__restrict__ v_all, int *__restrict__ not_prefix) {
  extern __shared__ float4 sall[];  // [m]: (x, y, z, -)
  const int b = blockIdx.y, part = threadIdx.x & (PFX_LANES - 1);
  const int i = blockIdx.x * (256 / PFX_LANES) + threadIdx.x / PFX_LANES;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *not_prefix = 0;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const int iend = min((int)(blockIdx.x + 1) * (256 / PFX_LANES), m);  // samples q < iend are all this block needs
  for (int q = threadIdx.x; q < iend; q += 256) sall[q] = make_float4(xyz[q * 3], xyz[q * 3 + 1], xyz[q * 3 + 2], 0.f);
  __syncthreads();
  const int ic = min(i, m - 1);
  const float4 me = sall[min(ic, iend - 1)];
  float v = 1e10f;
  for (int q = part; q < ic; q += PFX_LANES) {
    const float4 s = sall[q];
    v = fminf(v, vlp3d_sumsq3(me.x - s.x, me.y - s.y, me.z - s.z));
  }
#pragma unroll
  for (int off = 1; off < PFX_LANES; off <<= 1) v = fminf(v, __shfl_xor(v, off));
  if (i < m && part == 0) v_all[(size_t)b * m + i] = v;
}

__global__ __launch_bounds__(256) void fps_prefix_check8_kernel(const float *__restrict__ xyz_all, int N, int m,
                                                                const float *__restrict__ v_all, int *__restrict__ not_prefix) {
  extern __shared__ float4 sall[];  // [m - 1]: (x, y, z of sample q, v[q + 1])
  const int b = blockIdx.y, part = threadIdx.x & (PFX_LANES - 1);
  const int p = blockIdx.x * (256 / PFX_LANES) + threadIdx.x / PFX_LANES;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const float *__restrict__ v = v_all + (size_t)b * m;
  for (int q = threadIdx.x; q < m - 1; q += 256) sall[q] = make_float4(xyz[q * 3], xyz[q * 3 + 1], xyz[q * 3 + 2], v[q + 1]);
  __syncthreads();
  const int pc = min(p, N - 1);
  const float x = xyz[pc * 3], y = xyz[pc * 3 + 1], z = xyz[pc * 3 + 2];
  bool bad = false;
  if (part == 0) {
    bad = vlp3d_fps_skipped(x, y, z);
    if (p >= 1 && p < m) bad = bad || !(v[p] > 0.f);
  }
  const int steps = p < m ? p : m;
  const int nq = max(steps - 1, 0);                     // samples q = 0 .. nq - 1 carry a comparison (q + 1 < steps)
  const int len = (nq + PFX_LANES - 1) / PFX_LANES;
  const int q0 = min(part * len, nq), q1 = min(q0 + len, nq);
  float rmin = 1e10f;
  for (int q = q0; q < q1; ++q) {
    const float4 s = sall[q];
    rmin = fminf(rmin, vlp3d_sumsq3(x - s.x, y - s.y, z - s.z));
  }
  // exclusive prefix minimum over the group's lanes (lane ids inside the group are consecutive)
  float incl = rmin;
#pragma unroll
  for (int off = 1; off < PFX_LANES; off <<= 1) {
    const float o = __shfl_up(incl, off, PFX_LANES);
    if (part >= off) incl = fminf(incl, o);
  }
  float r = __shfl_up(incl, 1, PFX_LANES);
  if (part == 0) r = 1e10f;
  for (int q = q0; q < q1; ++q) {
    const float4 s = sall[q];
    r = fminf(r, vlp3d_sumsq3(x - s.x, y - s.y, z - s.z));
    bad = bad || !(r < s.w);
  }
  if (bad && p < N) *not_prefix = 1;
}
}  // namespace

// v: (B, m) floats of scratch; not_prefix: one int (0 = "the sampling order is 0..m-1" proven).  N, m <= 65536.
extern "C" int vlp3d_fps_prefix_check(const float *xyz, int B, int N, int m, float *v, int *not_prefix, void *stream) {
  if (!xyz || !v || !not_prefix || B < 1 || N < 1 || m < 1 || m > N || N > 65536) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (m <= 2048) {
    constexpr int PPB = 256 / PFX_LANES;  // points / samples per workgroup
    const size_t lds = (size_t)m * sizeof(float4);
    hipLaunchKernelGGL(fps_prefix_v8_kernel, dim3((m + PPB - 1) / PPB, B), dim3(256), lds, s, xyz, N, m, v, not_prefix);
    hipLaunchKernelGGL(fps_prefix_check8_kernel, dim3((N + PPB - 1) / PPB, B), dim3(256), lds, s, xyz, N, m, v, not_prefix);
  } else {
    hipLaunchKernelGGL(fps_prefix_v_kernel, dim3((m + 255) / 256, B), dim3(256), 0, s, xyz, N, m, v, not_prefix);
    hipLaunchKernelGGL(fps_prefix_check_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, xyz, N, m, v, not_prefix);
  }
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// vlp3d_furthest_point_sampling, skipped (idx = 0..m-1) when *not_prefix == 0 (vlp3d_fps_prefix_check ran before on the
// same stream); not_prefix == NULL: unconditional.
extern "C" int vlp3d_furthest_point_sampling_cond(const float *xyz, int B, int N, int m, float *temp, int *idx,
                                                  const int *not_prefix, void *stream) {
  return fps_dense(xyz, B, N, m, temp, idx, not_prefix, stream);
}
