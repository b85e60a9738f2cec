// Furthest point sampling with distance-bound pruning for gfx950 — same results, bit for bit, as the dense
// kernel of fps.hip (and therefore as the reference's sampling_gpu.cu:74-178), but most of the
// B*(m-1)*n distance updates are never executed.
//
// Idea: after j samples every running minimum temp[p] is <= r_j^2 (the current max-min distance), so the new
// sample q can only lower temp[p] for points within r_j of q.  Points are first sorted by a 15-bit Morton cell
// (counting sort, 4 small kernels), so that a "slot" = 64 consecutive sorted points (one per lane of one wave)
// is spatially compact.  Lane i of a wave keeps the state of the wave's slot i: its exact bounding box, its
// current maximum temp and that point's lane and coordinates.  Per iteration a lane tests its slot:
//     box_distance^2(q) * (1 - 1e-5)  <  slot max
// — if not, NO point of the slot can change (computed d >= box distance up to a few ulp, the margin covers fp32
// rounding) and the slot is skipped wholesale; its cached maximum still takes part in the argmax.  On the bench
// scenes ~14 of the 640 (slot, wave) pairs are active per iteration (2.2 %).  Active slots (ballot -> scalar loop)
// read their 64 points (x,y,z,temp as one float4; LDS for the first 9 slots of every wave, L2 for the rest),
// update temp, and recompute the slot maximum with one 32-bit wave reduction.
// Tie order (the reference's reduction tree prefers the smallest (bitrev_P(k mod P), k) among equal values), skip
// rule (|p|^2 <= 1e-3 -> never a candidate) and the fp32 distance expression are those of the dense kernel.
#include "common.h"
#include "fps_cells.h"

namespace {

using vlp3d_cells::BBOX_PARTS;
using vlp3d_cells::NCELL;

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
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_u64<0xB1>(v);
  v = dpp_max_u64<0x4E>(v);
  v = dpp_max_u64<0x141>(v);
  v = dpp_max_u64<0x140>(v);
  unsigned long long a = readlane_u64(v, 0), b = readlane_u64(v, 16);
  unsigned long long c = readlane_u64(v, 32), d = readlane_u64(v, 48);
  a = a > b ? a : b;
  c = c > d ? c : d;
  return a > c ? a : c;
}
template <bool IS_MAX>
__device__ __forceinline__ float wave_minmax_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = IS_MAX ? fmaxf(v, o) : fminf(v, o);
  }
  return v;
}
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ---- pre-pass 1: bounding boxes of BBOX_PARTS chunks of every scene, histogram cleared ------------------------
// (one workgroup per scene read its 480 KB alone: 16.5 us; the cell kernel folds the partial boxes itself)
__global__ __launch_bounds__(256) void fps_bbox_kernel(const float *__restrict__ xyz, int N, float *__restrict__ bbox,
                                                       int *__restrict__ hist) {
  __shared__ float red[6][4];
  const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
  const float *p = xyz + (size_t)b * N * 3;
  const int per = (N + BBOX_PARTS - 1) / BBOX_PARTS, k1 = min(N, (part + 1) * per);
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = part * per + tid; k < k1; k += 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float l = wave_minmax_f32<false>(lo[a]), h = wave_minmax_f32<true>(hi[a]);
    if ((tid & 63) == 0) {
      red[a][tid >> 6] = l;
      red[3 + a][tid >> 6] = h;
    }
  }
  for (int c = part * (NCELL / BBOX_PARTS) + tid; c < (part + 1) * (NCELL / BBOX_PARTS); c += 256) hist[(size_t)b * NCELL + c] = 0;
  __syncthreads();
  if (tid < 6) {
    float v = red[tid][0];
    for (int w = 1; w < 4; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
    bbox[(b * BBOX_PARTS + part) * 6 + tid] = v;
  }
}

// ---- pre-pass 2: Morton cell of every point + histogram ------------------------------------------------------
__global__ __launch_bounds__(256) void fps_cell_kernel(const float *__restrict__ xyz, int N,
                                                       const float *__restrict__ bbox, int *__restrict__ cellid,
                                                       int *__restrict__ hist, float *__restrict__ box) {
  __shared__ float bb[6];
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 6) {
    float v = bbox[b * BBOX_PARTS * 6 + threadIdx.x];
    for (int w = 1; w < BBOX_PARTS; ++w) {
      const float o = bbox[(b * BBOX_PARTS + w) * 6 + threadIdx.x];
      v = threadIdx.x < 3 ? fminf(v, o) : fmaxf(v, o);
    }
    bb[threadIdx.x] = v;
    if (blockIdx.x == 0) box[b * 8 + threadIdx.x] = v;  // the folded box, kept for the sorted ball query
  }
  __syncthreads();
  if (k >= N) return;
  const float *p = xyz + ((size_t)b * N + k) * 3;
  // 32 x 32 x 32 cells over the bounding box, numbered along the 3-D Hilbert curve (fps_cells.h: the sorted ball query of
  // csrc/ball_query_sorted.hip evaluates the same two functions for its cell ranges)
  const int cell = vlp3d_cells::cell_code((unsigned)vlp3d_cells::axis_cell(p[0], bb[0], bb[3]),
                                          (unsigned)vlp3d_cells::axis_cell(p[1], bb[1], bb[4]),
                                          (unsigned)vlp3d_cells::axis_cell(p[2], bb[2], bb[5]));
  // the point's RANK inside its cell comes back with the histogram update and travels with the cell code (15 + 17 bits, N <=
  // 131 072): the scatter below then needs no second round of 320 000 memory-side atomics (27 -> see DESIGN.md 4.1)
  const int rank = atomicAdd(hist + (size_t)b * NCELL + cell, 1);
  cellid[(size_t)b * N + k] = (int)((unsigned)cell | ((unsigned)rank << 15));
}

// ---- pre-pass 3: INCLUSIVE scan of the 32768 bins of a scene (in place): hist[c] = end of cell c's run -------
__global__ __launch_bounds__(1024) void fps_scan_kernel(int *__restrict__ hist) {
  __shared__ int part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  int *h = hist + (size_t)b * NCELL + tid * 32;
  int loc[32], s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    s += h[i];
    loc[i] = s;
  }
  // inclusive scan of the 1024 thread sums: shuffles inside a wave, the 16 wave totals through LDS (the Hillis-Steele form
  // over LDS took 20 barriers: 11.9 us for the 8 scenes, in front of the longest dependent chain of the step)
  const int lane = tid & 63, wave = tid >> 6;
  int inc = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) part[wave] = inc;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += part[w];
  const int base = before + inc - s;
#pragma unroll
  for (int i = 0; i < 32; ++i) h[i] = base + loc[i];
}

// ---- pre-pass 4: scatter into sorted order: pts = (x, y, z, temp0), perm = original index ---------------------
__global__ __launch_bounds__(256) void fps_scatter_kernel(const float *__restrict__ xyz, int N,
                                                          const int *__restrict__ cellid, int *__restrict__ hist,
                                                          float4 *__restrict__ pts, int *__restrict__ perm) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  const float *p = xyz + ((size_t)b * N + k) * 3;
  const float x = p[0], y = p[1], z = p[2];
  const unsigned v = (unsigned)cellid[(size_t)b * N + k];
  const int cell = (int)(v & (NCELL - 1)), rank = (int)(v >> 15);
  const int pos = (cell > 0 ? hist[(size_t)b * NCELL + cell - 1] : 0) + rank;
  pts[(size_t)b * N + pos] = make_float4(x, y, z, vlp3d_fps_skipped(x, y, z) ? -1.f : 1e10f);
  perm[(size_t)b * N + pos] = k;
}

// (Measured and rejected, round 3: the whole pre-pass of a scene in ONE workgroup — bounding box, cells kept in registers,
// the 32 768-bin histogram and the slot claims as LDS atomics, the scan in between — to replace the four launches' 320 000
// memory-side atomics each way and three kernel boundaries: 118-125 us against 81 us; forty Hilbert codes and two LDS atomics
// per thread on one CU per scene are slower than the same work spread over the chip.)
// ---- main kernel ------------------------------------------------------------------------------------------------
// One workgroup (16 waves) per scene.  The per-iteration dependency chain is what bounds it, so:
//  * the cached per-slot state is (max value, winner lane, winner coordinates); the reference's tie order among
//    EQUAL values is resolved exactly but lazily — only when a 32-bit max is attained more than once do the tied
//    lanes look up their original indices (perm) and compare the reference's tie key; a hierarchical max that breaks
//    ties by the same total order at every level returns the same winner as the flat 64-bit key reduction;
//  * reductions are 32-bit DPP reductions (the 64-bit form costs three times the instructions);
//  * the winner's COORDINATES travel with the candidate (slot -> wave -> block through LDS), so the next iteration
//    starts without a dependent global load of xyz[old]; idx[j] = perm[pos] is written off the critical path;
//  * the first L slots of every wave live in LDS (x,y,z,temp as float4, 16 KB per slot), the rest stays in L2.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {  // result is wave-uniform
  v = dpp_max_u32<0xB1>(v);
  v = dpp_max_u32<0x4E>(v);
  v = dpp_max_u32<0x141>(v);
  v = dpp_max_u32<0x140>(v);
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
// maximum over lanes 0..15 (one DPP row), valid in lane 0..15; the caller reads lane 0: one v_readlane instead of four + 3 s_max
__device__ __forceinline__ unsigned row0_max_u32(unsigned v) {
  v = dpp_max_u32<0xB1>(v);
  v = dpp_max_u32<0x4E>(v);
  v = dpp_max_u32<0x141>(v);
  v = dpp_max_u32<0x140>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}
__device__ __forceinline__ float readlane_f32(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// NS = slots per lane: a wave holds 64 * NS slots of 64 points, i.e. N <= 65536 * NS (NS = 2: cfg5's 80 000-point scenes).
// PROF: thread 0 of every workgroup accumulates s_memtime deltas of the iteration's phases into prof[blockIdx.x][0..5]
// (tools/fps_phases.py; the production instantiation PROF = false carries no trace of it).
template <int NS, bool PROF = false>
__global__ __launch_bounds__(1024) void fps_pruned_kernel(const float *__restrict__ xyz_all, float4 *__restrict__ pts_all,
                                                           const int *__restrict__ perm_all, int *__restrict__ idx_all,
                                                           int N, int m, int log2P, int L, int m_lds,
                                                           unsigned long long *__restrict__ prof = nullptr,
                                                           unsigned *__restrict__ trace = nullptr) {
  // The kernel is one dependent chain per scene on 8 of the 256 CUs while the step's dense kernels fill the chip: waves of
  // those kernels that land on the same SIMDs compete for instruction issue (3.4 ms inside the step against 2.8 ms alone).
  // Highest wave priority: the arbiter serves these waves first; the dense kernels lose nothing measurable.
  // (Tried and rejected, round 3: 16 more slots per wave in REGISTERS — 50 of the 128 VGPRs a 1024-thread workgroup may use
  // are taken — so that 25 instead of 9 of SA1's 40 slots per wave never touch L2.  A compare-and-select chain per access:
  // 2.80 -> 3.65 ms; indexed register moves (s_set_gpr_idx): 3.04 ms.  The L2 reads of an iteration's active slots are issued
  // together and are not what the slowest wave waits for.)
  __builtin_amdgcn_s_setprio(3);
#ifndef VLP3D_FPS_SHARE_CU
  // ... and the workgroup claims its CU's whole register file (16 waves x 128 VGPRs; the kernel itself needs 50): no wave of
  // another kernel can become resident on these 8 CUs while the chain runs, so nothing competes for their issue slots and LDS.
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
#endif
  extern __shared__ float4 lpts[];  // [L][1024]: slots 0..L-1 of every wave; then int s_out[m_lds]: sorted positions
  int *s_out = reinterpret_cast<int *>(lpts + (size_t)L * 1024);  // of the samples (idx = perm[pos], written at the end)
  // per parity and wave: the wave's candidate (value, sorted position) and its coordinates.  Lanes 0..15 fetch BOTH for
  // "their" wave right after the barrier (independent reads, in flight together); the winner's coordinates then come from
  // the winner LANE by v_readlane instead of a second, dependent LDS round trip (round 3: -150 cycles of the 670-cycle tail)
  __shared__ uint2 s_vp[2][16];
  __shared__ float4 s_xyz4[2][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  float4 *__restrict__ pts = pts_all + (size_t)b * N;
  const int *__restrict__ perm = perm_all + (size_t)b * N;
  int *__restrict__ idx = idx_all + (size_t)b * m;
  const int nslots = (N + 1023) / 1024;  // <= 64 NS: lane i of a wave holds the state of the wave's slots i, i + 64, ..
  const unsigned Pm1 = (1u << log2P) - 1u;

  // larger = preferred by the reference's reduction tree among equal values (see fps.hip)
  auto tiekey = [&](int orig) -> unsigned {
    const unsigned k = (unsigned)orig;
    return 0xFFFFFFFFu - (__brev(k & Pm1) | (k >> log2P));
  };
  auto value_of = [](float t, bool valid) -> unsigned { return (valid && t >= 0.f) ? __float_as_uint(t) + 1u : 0u; };
  // winner lane of a slot given every lane's value v (0 = not a candidate); vmax = wave max of v, > 0
  auto winner_lane = [&](unsigned v, unsigned vmax, int pos) -> int {
    const unsigned long long w = __ballot(v == vmax);
    if (__popcll(w) == 1) return __builtin_ctzll(w);
    const unsigned tk = (v == vmax) ? tiekey(perm[pos]) : 0u;  // exact tie: the reference's order decides
    const unsigned tmax = wave_max_u32(tk);
    return __builtin_ctzll(__ballot(v == vmax && tk == tmax));
  };

  float blo[NS][3], bhi[NS][3];
  unsigned sval[NS];
  int swl[NS];
  float sx[NS], sy[NS], sz[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    sval[q] = 0u; swl[q] = 0; sx[q] = sy[q] = sz[q] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) blo[q][a] = bhi[q][a] = 0.f;
  }
  for (int i = 0; i < nslots; ++i) {
    const int pos = i * 1024 + wave * 64 + lane;
    const bool valid = pos < N;
    float4 p = pts[valid ? pos : N - 1];
    if (!valid) p.w = -1.f;
    if (i < L) lpts[i * 1024 + tid] = p;
    float lo[3], hi[3];
    const float c[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = wave_minmax_f32<false>(valid ? c[a] : 3.0e38f);
      hi[a] = wave_minmax_f32<true>(valid ? c[a] : -3.0e38f);
    }
    const unsigned v = value_of(p.w, valid);
    const unsigned vmax = wave_max_u32(v);
    int wl = 0;
    if (vmax != 0u) wl = winner_lane(v, vmax, valid ? pos : N - 1);
    const float wx = readlane_f32(p.x, wl), wy = readlane_f32(p.y, wl), wz = readlane_f32(p.z, wl);
#pragma unroll
    for (int q = 0; q < NS; ++q)
      if (lane + 64 * q == i) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          blo[q][a] = lo[a];
          bhi[q][a] = hi[a];
        }
        sval[q] = vmax; swl[q] = wl; sx[q] = wx; sy[q] = wy; sz[q] = wz;
      }
  }
  if (tid == 0) idx[0] = 0;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  int par = 0;
  __syncthreads();

  unsigned long long ph[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tq = 0ull;
  unsigned tr[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // per-wave trace of one iteration: 5 phase deltas, LDS / global / re-reduced slots
#define FPS_MARK(k)                                        \
  if (PROF) {                                              \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ph[k] += now_ - tq;                                    \
    tr[k] = (unsigned)(now_ - tq);                         \
    tq = now_;                                             \
  }
  if (PROF) tq = __builtin_readcyclecounter();
  for (int j = 1; j < m; ++j) {
    if (PROF) tr[5] = tr[6] = tr[7] = 0u;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const float ex = fmaxf(0.f, fmaxf(blo[q][0] - x1, x1 - bhi[q][0]));
      const float ey = fmaxf(0.f, fmaxf(blo[q][1] - y1, y1 - bhi[q][1]));
      const float ez = fmaxf(0.f, fmaxf(blo[q][2] - z1, z1 - bhi[q][2]));
      const float lb2 = (ex * ex + ey * ey + ez * ez) * (1.0f - 1e-5f);
      const bool active = (lane + 64 * q < nslots) && (sval[q] != 0u) && (lb2 < __uint_as_float(sval[q] - 1u));
      unsigned long long todo = __ballot(active);
      FPS_MARK(0)  // bounding-box test of the cached slots + ballot
      while (todo != 0ull) {  // wave-uniform loop over this wave's active slots, four at a time
        int si[4];
        float4 p[4];
        int nb = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          si[u] = -1;
          if (todo != 0ull) {  // uniform
            si[u] = __builtin_ctzll(todo) + 64 * q;
            todo &= todo - 1ull;
            nb = u + 1;
            if (si[u] < L) {
              p[u] = lpts[si[u] * 1024 + tid];
            } else {
              // the barrier below does not wait for global stores any more: order this read after the wave's own
              // older temp stores to the slot (they were issued iterations ago; the wait is free in practice)
              __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) only (gfx9 encoding: expcnt/lgkmcnt fields left at max)
              const int pos = si[u] * 1024 + wave * 64 + lane;
              p[u] = pts[pos < N ? pos : N - 1];
              if (pos >= N) p[u].w = -1.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < nb) {  // uniform
            const int pos = si[u] * 1024 + wave * 64 + lane;
            const bool valid = pos < N;
            if (PROF) tr[si[u] < L ? 5 : 6] += 1u;
            const float d = vlp3d_sumsq3(p[u].x - x1, p[u].y - y1, p[u].z - z1);
            const float t = vmin(d, p[u].w);
            if (si[u] < L) lpts[si[u] * 1024 + tid].w = t;
            else if (valid) pts[pos].w = t;
            // The slot's cached maximum is still right unless the point that HOLDS it moved (values only decrease): only
            // then is the slot reduced again (round 3: the reduction + winner + three readlanes were ~2/3 of an update's
            // work, and the workgroup waits at the barrier for its slowest wave — tools/fps_phases.py).  Tried and rejected:
            // issuing the four slots of a batch as straight-line code with interleaved reductions (absent slots as
            // dummies): 2.79 -> 4.83 ms — most batches hold ONE slot, and 16 waves on one CU are bound by instruction issue,
            // not by the latency of a wave's own chain.
            const int wl_old = __builtin_amdgcn_readlane(swl[q], si[u] - 64 * q);
            if (__ballot(lane == wl_old && t < p[u].w) != 0ull) {  // wave-uniform
              if (PROF) tr[7] += 1u;
              const unsigned v = value_of(t, valid);
              const unsigned vmax = wave_max_u32(v);
              int wl = 0;
              if (vmax != 0u) wl = winner_lane(v, vmax, valid ? pos : N - 1);
              const float wx = readlane_f32(p[u].x, wl), wy = readlane_f32(p[u].y, wl), wz = readlane_f32(p[u].z, wl);
              if (lane + 64 * q == si[u]) { sval[q] = vmax; swl[q] = wl; sx[q] = wx; sy[q] = wy; sz[q] = wz; }
            }
          }
        }
      }
    }
    FPS_MARK(1)  // distance updates of the active slots (+ their slot maxima)
    // this lane's better slot (NS = 2): larger value; an exact tie is decided by the reference's order of the two winners
    unsigned lv = sval[0];
    int lsl = lane, lwl = swl[0];
    float lx = sx[0], ly = sy[0], lz = sz[0];
#pragma unroll
    for (int q = 1; q < NS; ++q) {
      bool take = sval[q] > lv;
      if (sval[q] == lv && lv != 0u && lane + 64 * q < nslots) {
        const int pa = min(lsl * 1024 + wave * 64 + lwl, N - 1), pb = min((lane + 64 * q) * 1024 + wave * 64 + swl[q], N - 1);
        take = tiekey(perm[pb]) > tiekey(perm[pa]);
      }
      if (take) { lv = sval[q]; lsl = lane + 64 * q; lwl = swl[q]; lx = sx[q]; ly = sy[q]; lz = sz[q]; }
    }
    // wave candidate = best slot (ties between slots by the reference's order of their winners)
    const unsigned mine = lsl < nslots ? lv : 0u;
    const unsigned vw = wave_max_u32(mine);
    int cl = 0;
    if (vw != 0u) cl = winner_lane(mine, vw, min(lsl * 1024 + wave * 64 + lwl, N - 1));
    const int cwl = __builtin_amdgcn_readlane(lwl, cl), csl = __builtin_amdgcn_readlane(lsl, cl);
    const float cx = readlane_f32(lx, cl), cy = readlane_f32(ly, cl), cz = readlane_f32(lz, cl);
    FPS_MARK(2)  // wave candidate: reduction over the lanes' cached slot maxima, winner's coordinates by readlane
    if (lane == 0) {
      s_vp[par][wave] = make_uint2(vw, (unsigned)(csl * 1024 + wave * 64 + cwl));
      s_xyz4[par][wave] = make_float4(cx, cy, cz, 0.f);
    }
    // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait ~1 us for the acknowledgement of the
    // temp stores of global slots and of the sample list — nobody else reads those before the kernel ends
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FPS_MARK(3)  // LDS write of the candidate + the workgroup barrier (= waiting for the slowest wave)
    // block winner, computed redundantly by every wave from the 16 candidates (no second barrier)
    const uint2 vp = s_vp[par][lane & 15];
    const float4 cxyz = s_xyz4[par][lane & 15];
    const unsigned bv = lane < 16 ? vp.x : 0u;
    const unsigned bpos = lane < 16 ? vp.y : 0u;
    const unsigned bmax = row0_max_u32(bv);  // the 16 candidates sit in lanes 0..15
    if (bmax != 0u) {
      const int wi = winner_lane(bv, bmax, (int)min(bpos, (unsigned)(N - 1)));
      x1 = readlane_f32(cxyz.x, wi); y1 = readlane_f32(cxyz.y, wi); z1 = readlane_f32(cxyz.z, wi);
      if (tid == 0) {
        const unsigned wpos = (unsigned)__builtin_amdgcn_readlane((int)vp.y, wi);
        if (m_lds) s_out[j] = (int)wpos;
        else idx[j] = perm[wpos];
      }
    } else {  // no candidate left (every point skipped): the reference returns index 0
      x1 = xyz[0]; y1 = xyz[1]; z1 = xyz[2];
      if (tid == 0) {
        if (m_lds) s_out[j] = -1;
        else idx[j] = 0;
      }
    }
    par ^= 1;
    FPS_MARK(4)  // block reduction over the 16 wave candidates, next sample's coordinates from LDS
    if (PROF && trace != nullptr && lane == 0) {
      uint4 *t4 = reinterpret_cast<uint4 *>(trace + (((size_t)b * m + j) * 16 + wave) * 8);
      t4[0] = make_uint4(tr[0], tr[1], tr[2], tr[3]);
      t4[1] = make_uint4(tr[4], tr[5], tr[6], tr[7]);
    }
  }
#undef FPS_MARK
  if (PROF && tid == 0)
    for (int k = 0; k < 5; ++k) prof[(size_t)b * 8 + k] = ph[k];
  if (m_lds) {  // sorted position -> original index, all threads
    __syncthreads();
    for (int j = 1 + tid; j < m; j += 1024) idx[j] = s_out[j] < 0 ? 0 : perm[s_out[j]];
  }
}

// ---- main kernel, round 4: running minima in REGISTERS, read-only coordinates ------------------------------------------------
// What the round-3 kernel paid for (tools/fps_trace.py, profiles/r04_fps_trace_*.txt): a slot update cost a wave ~740 (LDS
// slot) / ~930 (L2 slot) cycles + ~410 when the slot was reduced again, and beside a streaming load on the other 248 CUs the
// kernel ran 1.9x slower although its CU is its own: gfx9 returns vector-memory operations IN ORDER, so the load of a slot's
// points could not be consumed before the acknowledgement of the temp stores of earlier iterations — microseconds under load.
// Here the running minimum of every point lives in the register file: lane l of wave w keeps temp of point
// (slot * 1024 + w * 64 + l) in VGPR 64 + slot (a workgroup of 16 waves owns its CU's whole file anyway: 128 per lane; the
// compiler is held to v0..v63 by amdgpu_num_vgpr, v64..v127 are addressed with s_set_gpr_idx by the wave-uniform slot number —
// hipcc's own dynamic indexing of a 32-wide vector copies the whole vector at every control-flow join).  The loop issues NO
// store (the sample list stays in LDS), the coordinates are read-only: the first L slots of every wave from LDS as three
// dword arrays (12 KB per slot: 12 instead of 9 slots), the others from L2, a slot ahead of its use.  The update loop handles
// one slot per trip (the four-at-a-time form compiled to a dozen scalar branches per slot), and a wave whose slot maxima did
// not change re-publishes its cached candidate instead of reducing 64 lanes again.
__device__ __forceinline__ float fps_treg_read(int slot) {  // slot is wave-uniform, 0..63 (gfx9 has no v_movrel)
  float r;
  asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, v64\n\ts_set_gpr_idx_off" : "=v"(r) : "s"(slot));
  return r;
}
__device__ __forceinline__ void fps_treg_write(int slot, float v) {
  asm volatile("s_set_gpr_idx_on %1, gpr_idx(DST)\n\tv_mov_b32 v64, %0\n\ts_set_gpr_idx_off" ::"v"(v), "s"(slot));
}

// ABL (timing experiments of tools/fps_trace.py only; the outputs are then NOT the sampling): 1 = no slot is ever updated,
// 2 = a slot is never reduced again, 4 = the wave candidate is reduced every iteration
template <bool PROF, int ABL = 0>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(64))) void fps_pruned_reg_kernel(
    const float *__restrict__ xyz_all, const float4 *__restrict__ pts_all, const int *__restrict__ perm_all,
    int *__restrict__ idx_all, int N, int m, int log2P, int L, int m_lds, unsigned *__restrict__ trace) {
  __builtin_amdgcn_s_setprio(3);
  asm volatile("v_mov_b32 v127, 0" ::: "v127");  // the kernel descriptor claims 128 VGPRs: v64..v127 hold the running minima
  extern __shared__ float lxyz[];  // x[L][1024], y[L][1024], z[L][1024]; then int s_out[m_lds]
  int *s_out = reinterpret_cast<int *>(lxyz + (size_t)3 * L * 1024);
  __shared__ uint2 s_vp[2][16];
  __shared__ float4 s_xyz4[2][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const float4 *__restrict__ pts = pts_all + (size_t)b * N;
  const int *__restrict__ perm = perm_all + (size_t)b * N;
  int *__restrict__ idx = idx_all + (size_t)b * m;
  const int nslots = (N + 1023) / 1024;  // <= 64: lane i of a wave holds the state of the wave's slot i
  const unsigned Pm1 = (1u << log2P) - 1u;
  const int lstride = L * 1024;
  auto tiekey = [&](int orig) -> unsigned {
    const unsigned k = (unsigned)orig;
    return 0xFFFFFFFFu - (__brev(k & Pm1) | (k >> log2P));
  };
  auto value_of = [](float t) -> unsigned { return t >= 0.f ? __float_as_uint(t) + 1u : 0u; };  // absent points carry -1
  auto winner_lane = [&](unsigned v, unsigned vmax, int pos) -> int {
    const unsigned long long w = __ballot(v == vmax);
    if (__popcll(w) == 1) return __builtin_ctzll(w);
    const unsigned tk = (v == vmax) ? tiekey(perm[pos]) : 0u;  // exact tie: the reference's order decides
    const unsigned tmax = wave_max_u32(tk);
    return __builtin_ctzll(__ballot(v == vmax && tk == tmax));
  };

  float blo0 = 0.f, blo1 = 0.f, blo2 = 0.f, bhi0 = 0.f, bhi1 = 0.f, bhi2 = 0.f;
  unsigned sval = 0u;
  int swl = 0;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int i = 0; i < 64; ++i) {
    if (i >= nslots) {  // uniform
      fps_treg_write(i, -1.f);
      continue;
    }
    const int pos = i * 1024 + wave * 64 + lane;
    const bool valid = pos < N;
    float4 p = pts[valid ? pos : N - 1];
    if (!valid) p.w = -1.f;
    fps_treg_write(i, p.w);
    if (i < L) {
      lxyz[i * 1024 + tid] = p.x;
      lxyz[lstride + i * 1024 + tid] = p.y;
      lxyz[2 * lstride + i * 1024 + tid] = p.z;
    }
    const float lo0 = wave_minmax_f32<false>(valid ? p.x : 3.0e38f), hi0 = wave_minmax_f32<true>(valid ? p.x : -3.0e38f);
    const float lo1 = wave_minmax_f32<false>(valid ? p.y : 3.0e38f), hi1 = wave_minmax_f32<true>(valid ? p.y : -3.0e38f);
    const float lo2 = wave_minmax_f32<false>(valid ? p.z : 3.0e38f), hi2 = wave_minmax_f32<true>(valid ? p.z : -3.0e38f);
    const unsigned v = value_of(p.w);
    const unsigned vmax = wave_max_u32(v);
    int wl = 0;
    if (vmax != 0u) wl = winner_lane(v, vmax, valid ? pos : N - 1);
    const float wx = readlane_f32(p.x, wl), wy = readlane_f32(p.y, wl), wz = readlane_f32(p.z, wl);
    if (lane == i) {
      blo0 = lo0; blo1 = lo1; blo2 = lo2; bhi0 = hi0; bhi1 = hi1; bhi2 = hi2;
      sval = vmax; swl = wl; sx = wx; sy = wy; sz = wz;
    }
  }
  if (tid == 0) idx[0] = 0;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  int par = 0;
  __syncthreads();

  unsigned tr[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  unsigned long long tq = 0ull;
#define FPS_MARK(k)                                               \
  if (PROF) {                                                     \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    tr[k] = (unsigned)(now_ - tq);                                \
    tq = now_;                                                    \
  }
  if (PROF) tq = __builtin_readcyclecounter();
  // the wave's candidate, cached across iterations: value, sorted position, coordinates (all wave-uniform)
  unsigned c_v = 0u, c_pos = 0u;
  float c_x = 0.f, c_y = 0.f, c_z = 0.f;
  int c_slot = 0;
  bool changed = true;
  const int lane_off = wave * 64 + lane;
  for (int j = 1; j < m; ++j) {
    if (PROF) tr[5] = tr[6] = tr[7] = 0u;
    const float ex = fmaxf(0.f, fmaxf(blo0 - x1, x1 - bhi0));
    const float ey = fmaxf(0.f, fmaxf(blo1 - y1, y1 - bhi1));
    const float ez = fmaxf(0.f, fmaxf(blo2 - z1, z1 - bhi2));
    const float lb2 = (ex * ex + ey * ey + ez * ez) * (1.0f - 1e-5f);
    const bool active = (lane < nslots) && (sval != 0u) && (lb2 < __uint_as_float(sval - 1u));
    unsigned long long todo = __ballot(active);
    FPS_MARK(0)
    if ((ABL & 1) == 0 && todo != 0ull) {  // wave-uniform
      auto fetch = [&](int slot, float &px, float &py, float &pz) {
        if (slot < L) {  // uniform
          const int a = slot * 1024 + tid;
          px = lxyz[a]; py = lxyz[lstride + a]; pz = lxyz[2 * lstride + a];
        } else {
          const int pos = slot * 1024 + lane_off;
          const float4 g = pts[pos < N ? pos : N - 1];
          px = g.x; py = g.y; pz = g.z;
        }
      };
      int cur = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      float px, py, pz;
      fetch(cur, px, py, pz);
      for (;;) {
        int nxt = -1;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (todo != 0ull) {  // uniform: the next slot's coordinates are requested before this slot is worked on
          nxt = __builtin_ctzll(todo);
          todo &= todo - 1ull;
          fetch(nxt, qx, qy, qz);
        }
        if (PROF) tr[cur < L ? 5 : 6] += 1u;
        const float told = fps_treg_read(cur);
        const float d = vlp3d_sumsq3(px - x1, py - y1, pz - z1);
        const float t = vmin(d, told);
        fps_treg_write(cur, t);
        // the slot's cached maximum stays valid unless the point that HOLDS it moved (values only decrease)
        const int wl_old = __builtin_amdgcn_readlane(swl, cur);
        if ((ABL & 2) == 0 && __ballot(lane == wl_old && t < told) != 0ull) {  // wave-uniform
          if (PROF) tr[7] += 1u;
          const unsigned v = value_of(t);
          const unsigned vmax = wave_max_u32(v);
          int wl = 0;
          if (vmax != 0u) wl = winner_lane(v, vmax, min(cur * 1024 + lane_off, N - 1));
          const float wx = readlane_f32(px, wl), wy = readlane_f32(py, wl), wz = readlane_f32(pz, wl);
          if (lane == cur) { sval = vmax; swl = wl; sx = wx; sy = wy; sz = wz; }
          // the wave's candidate can only change when the slot that HOLDS it was reduced again: another slot's new maximum is
          // <= its old one, and among equal values the old candidate already preceded every point of that slot in the
          // reference's order (it preceded the slot's old holder, which preceded the slot's other points)
          if (cur == c_slot) changed = true;
        }
        if (nxt < 0) break;
        cur = nxt; px = qx; py = qy; pz = qz;
      }
    }
    FPS_MARK(1)
    if (changed || (ABL & 4)) {  // wave-uniform: the maximum of the candidate's slot changed (or first iteration)
      const unsigned mine = lane < nslots ? sval : 0u;
      c_v = wave_max_u32(mine);
      int cl = 0;
      if (c_v != 0u) cl = winner_lane(mine, c_v, min(lane * 1024 + wave * 64 + swl, N - 1));
      c_slot = cl;
      c_pos = (unsigned)(cl * 1024 + wave * 64 + __builtin_amdgcn_readlane(swl, cl));
      c_x = readlane_f32(sx, cl); c_y = readlane_f32(sy, cl); c_z = readlane_f32(sz, cl);
      changed = false;
    }
    FPS_MARK(2)
    if (lane == 0) {
      s_vp[par][wave] = make_uint2(c_v, c_pos);
      s_xyz4[par][wave] = make_float4(c_x, c_y, c_z, 0.f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FPS_MARK(3)
    const uint2 vp = s_vp[par][lane & 15];
    const float4 cxyz = s_xyz4[par][lane & 15];
    const unsigned bv = lane < 16 ? vp.x : 0u;
    const unsigned bpos = lane < 16 ? vp.y : 0u;
    const unsigned bmax = row0_max_u32(bv);
    if (bmax != 0u) {
      const int wi = winner_lane(bv, bmax, (int)min(bpos, (unsigned)(N - 1)));
      x1 = readlane_f32(cxyz.x, wi); y1 = readlane_f32(cxyz.y, wi); z1 = readlane_f32(cxyz.z, wi);
      if (tid == 0) {
        const unsigned wpos = (unsigned)__builtin_amdgcn_readlane((int)vp.y, wi);
        if (m_lds) s_out[j] = (int)wpos;
        else idx[j] = perm[wpos];
      }
    } else {  // no candidate left (every point skipped): the reference returns index 0
      x1 = xyz[0]; y1 = xyz[1]; z1 = xyz[2];
      if (tid == 0) {
        if (m_lds) s_out[j] = -1;
        else idx[j] = 0;
      }
    }
    par ^= 1;
    FPS_MARK(4)
    if (PROF && trace != nullptr && lane == 0) {
      uint4 *t4 = reinterpret_cast<uint4 *>(trace + (((size_t)b * m + j) * 16 + wave) * 8);
      t4[0] = make_uint4(tr[0], tr[1], tr[2], tr[3]);
      t4[1] = make_uint4(tr[4], tr[5], tr[6], tr[7]);
    }
  }
#undef FPS_MARK
  if (m_lds) {
    __syncthreads();
    for (int j = 1 + tid; j < m; j += 1024) idx[j] = s_out[j] < 0 ? 0 : perm[s_out[j]];
  }
}

int reference_log2_block(int n) {
  int p = (int)(log((double)n) / log(2.0));
  if (p > 9) p = 9;
  if (p < 0) p = 0;
  return p;
}

}  // namespace

extern "C" long long vlp3d_fps_workspace_bytes(int B, int N) {
  if (B < 1 || N < 1) return 0;
  return vlp3d_cells::workspace_bytes(B, N);
}

// Pruned FPS.  workspace: vlp3d_fps_workspace_bytes(B, N) bytes of device scratch (contents ignored / clobbered).
// Requires N <= 131072 (64 slots per wave and lane-slot; two lane-slots above 65536); same output as
// vlp3d_furthest_point_sampling.
// variant: 0 = the register-resident kernel (N <= 65536; round 4), 1 = the round-3 kernel (temps in LDS / L2; always above 65536)
static int fps_pruned_launch(const float *xyz, int B, int N, int m, void *workspace, long long workspace_bytes, int *idx,
                             unsigned long long *prof, void *stream, unsigned *trace = nullptr, int lds_slots = -1,
                             int variant = 0);

extern "C" int vlp3d_furthest_point_sampling_pruned(const float *xyz, int B, int N, int m, void *workspace,
                                                    long long workspace_bytes, int *idx, void *stream) {
  return fps_pruned_launch(xyz, B, N, m, workspace, workspace_bytes, idx, nullptr, stream);
}

// Same sampling with the main kernel's per-phase cycle counts: phases (B x 8) u64, entries 0..4 of scene b = shader-clock
// cycles thread 0 spent in [slot test | slot updates | wave candidate | LDS + barrier | block reduction] summed over the
// m - 1 iterations (tools/fps_phases.py).  N <= 65536.
extern "C" int vlp3d_fps_pruned_profile(const float *xyz, int B, int N, int m, void *workspace, long long workspace_bytes,
                                        int *idx, unsigned long long *phases, void *stream) {
  if (!phases || N > 65536) return VLP3D_EINVAL;
  return fps_pruned_launch(xyz, B, N, m, workspace, workspace_bytes, idx, phases, stream, nullptr, -1, 1);  // round-3 kernel
}

// Diagnostic form (tools/fps_trace.py): variant 0 = register-resident kernel, 1 = round-3 kernel; lds_slots >= 0 overrides
// the number of slots per wave whose points live in LDS (0..12 / 0..9); with `phases` the profiling instantiation runs (variant
// 0 fills `phases` only through `trace`) and, when `trace` is given too, lane 0 of every wave writes 8 words per iteration — trace[((b*m + j)*16 + wave)*8 + k]: shader-clock cycles of the five phases, then the number of LDS-resident /
// L2-resident slots the wave updated and how many of them were reduced again.  N <= 65536 with phases.
extern "C" int vlp3d_fps_pruned_trace(const float *xyz, int B, int N, int m, void *workspace, long long workspace_bytes,
                                      int *idx, unsigned long long *phases, unsigned *trace, int lds_slots, int variant,
                                      void *stream) {
  if ((phases && N > 65536) || (trace && !phases) || lds_slots > 12 || variant < 0 || (variant > 1 && variant < 16) || (variant != 1 && N > 65536))
    return VLP3D_EINVAL;
  return fps_pruned_launch(xyz, B, N, m, workspace, workspace_bytes, idx, phases, stream, trace, lds_slots, variant);
}

static int fps_pruned_launch(const float *xyz, int B, int N, int m, void *workspace, long long workspace_bytes, int *idx,
                             unsigned long long *prof, void *stream, unsigned *trace, int lds_slots, int variant) {
  if (!xyz || !workspace || !idx || B < 1 || N < 1 || N > 131072 || m < 0 ||
      workspace_bytes < vlp3d_fps_workspace_bytes(B, N))
    return VLP3D_EINVAL;
  if (m == 0) return VLP3D_OK;
  hipStream_t s = (hipStream_t)stream;
  const vlp3d_cells::Workspace ws = vlp3d_cells::workspace_layout(workspace, B, N);
  float4 *pts = ws.pts;
  int *perm = ws.perm, *cellid = ws.cellid, *hist = ws.hist;
  float *bbox = ws.bbox;
  const dim3 gridN((N + 255) / 256, B);
  hipLaunchKernelGGL(fps_bbox_kernel, dim3(BBOX_PARTS, B), dim3(256), 0, s, xyz, N, bbox, hist);
  hipLaunchKernelGGL(fps_cell_kernel, gridN, dim3(256), 0, s, xyz, N, bbox, cellid, hist, ws.box);
  hipLaunchKernelGGL(fps_scan_kernel, dim3(B), dim3(1024), 0, s, hist);
  hipLaunchKernelGGL(fps_scatter_kernel, gridN, dim3(256), 0, s, xyz, N, cellid, hist, pts, perm);
  const int nslots = (N + 1023) / 1024;
  const int m_lds = m <= 2048 ? m : 0;  // the sample list is collected in LDS (8 KB) and written out at the end
  const int log2P = reference_log2_block(N);
  if (nslots > 64 && variant == 0) variant = 1;
  if (variant == 0) {
    int L = nslots < 12 ? nslots : 12;  // 12 x 12 KB of the 160 KB LDS hold the coordinates of slots 0..11 of every wave
    if (lds_slots >= 0 && lds_slots < L) L = lds_slots;
    const size_t lds = (size_t)L * 1024 * 12 + (size_t)m_lds * sizeof(int);
    const int max_lds = 12 * 1024 * 12 + 2048 * (int)sizeof(int);
    static std::atomic<unsigned long long> done_p{0}, done_r{0};
    const int e = prof ? vlp3d_opt_in_lds((const void *)fps_pruned_reg_kernel<true>, max_lds, done_p)
                       : vlp3d_opt_in_lds((const void *)fps_pruned_reg_kernel<false>, max_lds, done_r);
    if (e != VLP3D_OK) return e;
    if (prof)
      hipLaunchKernelGGL(fps_pruned_reg_kernel<true>, dim3(B), dim3(1024), lds, s, xyz, pts, perm, idx, N, m, log2P, L, m_lds, trace);
    else
      hipLaunchKernelGGL(fps_pruned_reg_kernel<false>, dim3(B), dim3(1024), lds, s, xyz, pts, perm, idx, N, m, log2P, L, m_lds,
                         (unsigned *)nullptr);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  if (variant >= 16) {  // timing experiments (tools/fps_trace.py): the kernel with a part of its iteration left out
    if (nslots > 64 || prof) return VLP3D_EINVAL;
    const int L = nslots < 12 ? nslots : 12;
    const size_t lds = (size_t)L * 1024 * 12 + (size_t)m_lds * sizeof(int);
    const int max_lds = 12 * 1024 * 12 + 2048 * (int)sizeof(int);
    static std::atomic<unsigned long long> d1{0}, d2{0}, d4{0};
#define FPS_ABL(A, D)                                                                                                          \
  case 16 + A: {                                                                                                               \
    const int e = vlp3d_opt_in_lds((const void *)fps_pruned_reg_kernel<false, A>, max_lds, D);                                 \
    if (e != VLP3D_OK) return e;                                                                                               \
    hipLaunchKernelGGL((fps_pruned_reg_kernel<false, A>), dim3(B), dim3(1024), lds, s, xyz, pts, perm, idx, N, m, log2P, L, m_lds, \
                       (unsigned *)nullptr);                                                                                   \
  } break;
    switch (variant) {
      FPS_ABL(1, d1)
      FPS_ABL(2, d2)
      FPS_ABL(4, d4)
      default: return VLP3D_EINVAL;
    }
#undef FPS_ABL
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  int L = nslots < 9 ? nslots : 9;  // 9 x 16 KB of the 160 KB LDS hold slots 0..8 of every wave
  if (lds_slots >= 0 && lds_slots < L) L = lds_slots;
  const size_t lds = (size_t)L * 1024 * sizeof(float4) + (size_t)m_lds * sizeof(int);
  const int max_lds = 9 * 1024 * (int)sizeof(float4) + 2048 * (int)sizeof(int);
  static std::atomic<unsigned long long> done1{0}, done2{0}, done1p{0};
  int e = vlp3d_opt_in_lds((const void *)fps_pruned_kernel<1>, max_lds, done1);
  if (e == VLP3D_OK) e = vlp3d_opt_in_lds((const void *)fps_pruned_kernel<2>, max_lds, done2);
  if (e == VLP3D_OK) e = vlp3d_opt_in_lds((const void *)fps_pruned_kernel<1, true>, max_lds, done1p);
  if (e != VLP3D_OK) return e;
  if (prof)
    hipLaunchKernelGGL((fps_pruned_kernel<1, true>), dim3(B), dim3(1024), lds, s, xyz, pts, perm, idx, N, m, log2P, L, m_lds, prof,
                       trace);
  else if (nslots <= 64)
    hipLaunchKernelGGL(fps_pruned_kernel<1>, dim3(B), dim3(1024), lds, s, xyz, pts, perm, idx, N, m, log2P, L, m_lds);
  else
    hipLaunchKernelGGL(fps_pruned_kernel<2>, dim3(B), dim3(1024), lds, s, xyz, pts, perm, idx, N, m, log2P, L, m_lds);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
