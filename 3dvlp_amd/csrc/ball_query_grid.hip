// Ball query through a uniform grid — same output, bit for bit, as csrc/ball_query.hip (and therefore as
// ball_query_gpu.cu:14-59 under the documented fp32 evaluation order), without testing every point against every centre.
//
// A ball of radius r around a centre only contains points of the 27 grid cells around the centre's cell when the cell
// edge is >= r.  Per call (all kernels on the caller's stream, workspace caller-owned):
//   bq_bbox /   scene bounding box (32 workgroups per scene, then one wave) -> grid origin / dimensions
//   bq_header   (cell edge = r * 1.001, enlarged until the grid has at
//               most MAX_CELLS cells); the 0.1 % margin covers fp32 rounding of the cell index AND of the distance test:
//               a point with computed d^2 < r^2 is at most r (1 + 1e-6) away.
//   bq_count    cell of every point (kept), histogram of the cells (integer atomics)
//   bq_scan     exclusive scan of the histogram per scene (one workgroup per scene)
//   bq_scatter  (x, y, z, index) of every point into its cell's segment (order inside a cell arbitrary)
//   bq_query    one wave per centre: the 9 (y, z) rows of its 3 x 3 x 3 neighbourhood are 9 CONTIGUOUS runs of the sorted
//               array (cells of equal (y, z) are adjacent in x); 64 lanes test 64 candidates per step with the SAME
//               distance expression as the brute-force kernel and append the hits to an LDS list by ballot / popcount;
//               the list is then ranked (rank = number of hits with a smaller index) and the nsample smallest indices are
//               written in ascending order, padded with the smallest — exactly what the ordered scan over all points
//               produces.  A neighbourhood with more than MAX_HITS hits falls back to that scan for this one centre.
// SA1 of cfg2 (8 x 2048 centres, 40 000 points, r = 0.2, 64 samples): 655 M distance tests become ~4 M.
#include "common.h"

namespace {

constexpr int MAX_CELLS = 1 << 17;  // per scene
constexpr int MAX_HITS = 1024;      // LDS list of one centre (4 KB per wave)
constexpr int START_LD = MAX_CELLS + 4;  // per-scene stride of the start array (ncell + 1 entries, rows 16-byte aligned)

struct GridHdr {  // one per scene, 16 floats / ints
  float ox, oy, oz, inv_cell;
  int nx, ny, nz, ncell;
};

__device__ __forceinline__ int cell_coord(float p, float o, float inv, int n) {
  int c = (int)floorf((p - o) * inv);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

constexpr int BBOX_PARTS = 32;  // workgroups per scene of the bounding-box pass

// partial[b][part][6] = min xyz, max xyz of this workgroup's slice of scene b.  The same workgroups clear the scene's cell
// counters (MAX_CELLS / BBOX_PARTS = 4096 each) — in-stream, not a hipMemsetAsync (common.h: vlp3d_zero_words explains why).
__global__ __launch_bounds__(256) void bq_bbox_kernel(const float *__restrict__ xyz, int N, float *__restrict__ partial,
                                                      int *__restrict__ count) {
  __shared__ float red[6][4];
  const int b = blockIdx.y, part = blockIdx.x;
  static_assert(MAX_CELLS % (BBOX_PARTS * 1024) == 0, "counter slice of a workgroup: whole int4 rounds");
  int4 *cz = reinterpret_cast<int4 *>(count + (size_t)b * MAX_CELLS + (size_t)part * (MAX_CELLS / BBOX_PARTS));
#pragma unroll
  for (int j = 0; j < MAX_CELLS / BBOX_PARTS / 1024; ++j) cz[j * 256 + threadIdx.x] = make_int4(0, 0, 0, 0);
  const float *p = xyz + (size_t)b * N * 3;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = part * 256 + threadIdx.x; i < N; i += BBOX_PARTS * 256)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = p[i * 3 + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (int off = 32; off >= 1; off >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], off));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off));
    }
    if ((threadIdx.x & 63) == 0) {
      red[c][threadIdx.x >> 6] = mn[c];
      red[3 + c][threadIdx.x >> 6] = mx[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const float *r = red[threadIdx.x];
    partial[((size_t)b * BBOX_PARTS + part) * 6 + threadIdx.x] =
        threadIdx.x < 3 ? fminf(fminf(r[0], r[1]), fminf(r[2], r[3])) : fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
  }
}

// grid header of scene b from the partial boxes (one workgroup of 64 threads per scene)
__global__ __launch_bounds__(64) void bq_header_kernel(const float *__restrict__ partial, float radius, GridHdr *__restrict__ hdr) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float v[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float x = lane < BBOX_PARTS ? partial[((size_t)b * BBOX_PARTS + lane) * 6 + c] : (c < 3 ? 3.0e38f : -3.0e38f);
    for (int off = 32; off >= 1; off >>= 1) x = c < 3 ? fminf(x, __shfl_xor(x, off)) : fmaxf(x, __shfl_xor(x, off));
    v[c] = x;
  }
  if (lane == 0) {
    float cell = radius * 1.001f;
    int nx, ny, nz;
    for (;;) {  // enlarge the cells until the grid is small enough (a larger cell is still correct, only less selective)
      nx = (int)((v[3] - v[0]) / cell) + 1;
      ny = (int)((v[4] - v[1]) / cell) + 1;
      nz = (int)((v[5] - v[2]) / cell) + 1;
      if ((long long)nx * ny * nz <= MAX_CELLS) break;
      cell *= 1.26f;
    }
    GridHdr h = {v[0], v[1], v[2], 1.f / cell, nx, ny, nz, nx * ny * nz};
    hdr[b] = h;
  }
}

__global__ __launch_bounds__(256) void bq_count_kernel(const float *__restrict__ xyz, int B, int N,
                                                       const GridHdr *__restrict__ hdr, int *__restrict__ cell_of,
                                                       int *__restrict__ count) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * N) return;
  const int b = (int)(t / N);
  const GridHdr h = hdr[b];
  const float *p = xyz + t * 3;
  const int c = (cell_coord(p[2], h.oz, h.inv_cell, h.nz) * h.ny + cell_coord(p[1], h.oy, h.inv_cell, h.ny)) * h.nx +
                cell_coord(p[0], h.ox, h.inv_cell, h.nx);
  cell_of[t] = c;
  atomicAdd(count + (size_t)b * MAX_CELLS + c, 1);
}

// start[b][c] = exclusive prefix of count[b][.] (start has ncell + 1 entries); cursor = copy of start.  One workgroup per
// scene walks the cells 1024 at a time (coalesced): shuffle scan inside each wave, the 16 wave totals scanned by wave 0.
__global__ __launch_bounds__(1024) void bq_scan_kernel(const GridHdr *__restrict__ hdr, const int *__restrict__ count,
                                                       int *__restrict__ start, int *__restrict__ cursor) {
  __shared__ int wtot[16];
  __shared__ int carry_s;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncell = hdr[b].ncell;
  const int *cnt = count + (size_t)b * MAX_CELLS;
  int *st = start + (size_t)b * START_LD, *cu = cursor + (size_t)b * MAX_CELLS;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < ncell; c0 += 4096) {  // four consecutive cells per thread (16-byte loads / stores)
    const int i = c0 + 4 * threadIdx.x;
    int4 v = make_int4(0, 0, 0, 0);
    if (i + 3 < ncell) v = *reinterpret_cast<const int4 *>(cnt + i);
    else {
      if (i < ncell) v.x = cnt[i];
      if (i + 1 < ncell) v.y = cnt[i + 1];
      if (i + 2 < ncell) v.z = cnt[i + 2];
    }
    const int mine = v.x + v.y + v.z + v.w;
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int w = lane < 16 ? wtot[lane] : 0;  // every wave scans the 16 wave totals itself (no second barrier round)
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(w, off);
      if (lane >= off) w += t;
    }
    const int before = wave > 0 ? __shfl(w, wave - 1) : 0, all = __shfl(w, 15);
    const int base = carry_s + before + inc - mine;
    const int4 o = make_int4(base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z);
    if (i + 3 < ncell) {
      *reinterpret_cast<int4 *>(st + i) = o;
      *reinterpret_cast<int4 *>(cu + i) = o;
    } else {
      if (i < ncell) { st[i] = o.x; cu[i] = o.x; }
      if (i + 1 < ncell) { st[i + 1] = o.y; cu[i + 1] = o.y; }
      if (i + 2 < ncell) { st[i + 2] = o.z; cu[i + 2] = o.z; }
    }
    __syncthreads();  // every wave has read carry_s and wtot
    if (threadIdx.x == 0) carry_s += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) st[ncell] = carry_s;
}

__global__ __launch_bounds__(256) void bq_scatter_kernel(const float *__restrict__ xyz, int B, int N,
                                                         const int *__restrict__ cell_of, int *__restrict__ cursor,
                                                         float4 *__restrict__ sorted) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * N) return;
  const int b = (int)(t / N), k = (int)(t - (long long)b * N);
  const int pos = atomicAdd(cursor + (size_t)b * MAX_CELLS + cell_of[t], 1);
  const float *p = xyz + t * 3;
  sorted[(size_t)b * N + pos] = make_float4(p[0], p[1], p[2], __int_as_float(k));
}

__global__ __launch_bounds__(256) void bq_query_kernel(const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all,
                                                       const GridHdr *__restrict__ hdr, const int *__restrict__ start_all,
                                                       const float4 *__restrict__ sorted_all, int *__restrict__ idx_all, int B,
                                                       int N, int M, float radius2, int nsample) {
  __shared__ int s_hits[4][MAX_HITS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long t = (long long)blockIdx.x * 4 + wave;
  if (t >= (long long)B * M) return;  // wave-uniform; no workgroup barrier below
  const int b = (int)(t / M);
  const GridHdr h = hdr[b];
  const float *c3 = new_xyz_all + t * 3;
  const float cx = c3[0], cy = c3[1], cz = c3[2];
  const int gx = cell_coord(cx, h.ox, h.inv_cell, h.nx), gy = cell_coord(cy, h.oy, h.inv_cell, h.ny),
            gz = cell_coord(cz, h.oz, h.inv_cell, h.nz);
  const int *start = start_all + (size_t)b * START_LD;
  const float4 *sorted = sorted_all + (size_t)b * N;
  int *hits = s_hits[wave];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int cnt = 0;
  const int x0 = max(gx - 1, 0), x1 = min(gx + 1, h.nx - 1);
  // The nine (y, z) rows of the neighbourhood are nine contiguous runs of the sorted array.  Lanes 0..8 fetch their run's
  // bounds at once, a wave scan turns the lengths into offsets of ONE candidate list, and the list is then walked 64
  // candidates per step with independent loads (walking the runs one after the other cost two dependent memory
  // latencies per run: 13 us per centre).
  int rs = 0, rl = 0;  // this lane's run: start, length (lanes >= 9 or rows outside the grid: empty)
  if (lane < 9) {
    const int z = gz + lane / 3 - 1, y = gy + lane % 3 - 1;
    if (z >= 0 && z < h.nz && y >= 0 && y < h.ny) {
      const int row = (z * h.ny + y) * h.nx;
      rs = start[row + x0];
      rl = start[row + x1 + 1] - rs;
    }
  }
  int pre = rl;  // inclusive scan over lanes 0..15 (only 0..8 are non-zero)
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
    const int v = __shfl_up(pre, off);
    if ((lane & 15) >= off) pre += v;
  }
  const int total = __shfl(pre, 8);
  int r_start[9], r_end[9];  // wave-uniform copies: global start of run j, end offset of run j in the candidate list
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    r_start[j] = __shfl(rs, j);
    r_end[j] = __shfl(pre, j);
  }
  auto cand = [&](int c) -> int {  // position in the sorted array of candidate c (c < total)
    int run = 0, base = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c >= r_end[j]) { run = j + 1; base = r_end[j]; }
    int gs = r_start[0];
#pragma unroll
    for (int j = 1; j < 9; ++j) gs = run == j ? r_start[j] : gs;
    return gs + (c - base);
  };
  for (int c0 = 0; c0 < total; c0 += 256) {  // four chunks of 64 candidates requested together
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 64 * u + lane;
      p[u] = sorted[cand(c < total ? c : total - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 64 * u + lane;
      const bool hit = c < total && vlp3d_sumsq3(cx - p[u].x, cy - p[u].y, cz - p[u].z) < radius2;  // the scan kernel's expression
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull) {
        const int pos = cnt + __popcll(mask & lt_mask);
        if (hit && pos < MAX_HITS) hits[pos] = __float_as_int(p[u].w);
        cnt += __popcll(mask);
      }
    }
  }
  int *row_out = idx_all + t * nsample;
  if (cnt > MAX_HITS) {  // pathological density: the ordered scan over all points, for this centre only
    const float *xyz = xyz_all + (size_t)b * N * 3;
    int n = 0, first = 0;
    for (int k0 = 0; k0 < N && n < nsample; k0 += 64) {
      const int k = k0 + lane;
      const bool hit = k < N && vlp3d_sumsq3(cx - xyz[k * 3], cy - xyz[k * 3 + 1], cz - xyz[k * 3 + 2]) < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull) {
        if (n == 0) first = k0 + __ffsll((long long)mask) - 1;
        const int pos = n + __popcll(mask & lt_mask);
        if (hit && pos < nsample) row_out[pos] = k;
        n += __popcll(mask);
      }
    }
    n = n < nsample ? n : nsample;
    for (int l = n + lane; l < nsample; l += 64) row_out[l] = first;
    return;
  }
  if (cnt <= 64) {  // the common case: one hit per lane, ranked through cross-lane reads (no LDS round trips)
    const int me = lane < cnt ? hits[lane] : 0x7fffffff;
    int rk = 0;
    for (int i = 0; i < cnt; ++i) rk += __shfl(me, i) < me ? 1 : 0;
    if (lane < cnt && rk < nsample) row_out[rk] = me;
    int smallest = me;
    for (int off = 32; off >= 1; off >>= 1) smallest = min(smallest, __shfl_xor(smallest, off));
    const int n = cnt < nsample ? cnt : nsample;
    const int pad = cnt > 0 ? smallest : 0;
    for (int l = n + lane; l < nsample; l += 64) row_out[l] = pad;
    return;
  }
  // rank every hit (number of hits with a smaller index; indices are distinct) and emit the nsample smallest in order
  int mine[MAX_HITS / 64], rank[MAX_HITS / 64];
#pragma unroll
  for (int j = 0; j < MAX_HITS / 64; ++j) {
    mine[j] = (j * 64 + lane < cnt) ? hits[j * 64 + lane] : 0x7fffffff;
    rank[j] = 0;
  }
  const int nj = (cnt + 63) / 64;  // wave-uniform
  for (int i = 0; i < cnt; ++i) {
    const int v = hits[i];  // LDS broadcast
#pragma unroll
    for (int j = 0; j < MAX_HITS / 64; ++j)
      if (j < nj) rank[j] += v < mine[j] ? 1 : 0;
  }
  int smallest = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < MAX_HITS / 64; ++j) {
    if (j < nj && j * 64 + lane < cnt) {
      if (rank[j] < nsample) row_out[rank[j]] = mine[j];
      if (rank[j] == 0) smallest = mine[j];
    }
  }
  for (int off = 32; off >= 1; off >>= 1) smallest = min(smallest, __shfl_xor(smallest, off));
  const int n = cnt < nsample ? cnt : nsample;
  const int pad = cnt > 0 ? smallest : 0;  // empty ball: an all-zero row, like the reference's pre-zeroed output
  for (int l = n + lane; l < nsample; l += 64) row_out[l] = pad;
}

}  // namespace

extern "C" long long vlp3d_ball_query_grid_workspace_bytes(int B, int N) {
  if (B < 1 || N < 1) return 0;
  // headers | cell of each point | count | start (+1) | cursor | sorted (x,y,z,idx), each 16-byte aligned
  long long bytes = (long long)B * 32 + (long long)B * BBOX_PARTS * 6 * 4;
  bytes += ((long long)B * N * 4 + 15) / 16 * 16;
  bytes += (long long)B * MAX_CELLS * 4;
  bytes += (long long)B * START_LD * 4;
  bytes += (long long)B * MAX_CELLS * 4;
  bytes += (long long)B * N * 16;
  return bytes;
}

extern "C" int vlp3d_ball_query_grid(const float *new_xyz, const float *xyz, int B, int N, int M, float radius, int nsample,
                                     void *workspace, long long workspace_bytes, int *idx, void *stream) {
  if (!new_xyz || !xyz || !idx || !workspace || B < 1 || N < 1 || M < 1 || nsample < 1 || !(radius > 0.f)) return VLP3D_EINVAL;
  if ((long long)N * 3 >= (1ll << 31) || (long long)B * N >= (1ll << 31) || (long long)B * M >= (1ll << 31)) return VLP3D_EINVAL;
  if (workspace_bytes < vlp3d_ball_query_grid_workspace_bytes(B, N) || ((size_t)workspace & 15)) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  char *w = (char *)workspace;
  GridHdr *hdr = (GridHdr *)w;
  w += (long long)B * 32;
  float *partial = (float *)w;
  w += (long long)B * BBOX_PARTS * 6 * 4;
  int *cell_of = (int *)w;
  w += ((long long)B * N * 4 + 15) / 16 * 16;
  int *count = (int *)w;
  w += (long long)B * MAX_CELLS * 4;
  int *start = (int *)w;
  w += (long long)B * START_LD * 4;
  int *cursor = (int *)w;
  w += (long long)B * MAX_CELLS * 4;
  float4 *sorted = (float4 *)w;
  const unsigned pblocks = (unsigned)(((long long)B * N + 255) / 256);
  hipLaunchKernelGGL(bq_bbox_kernel, dim3(BBOX_PARTS, B), dim3(256), 0, s, xyz, N, partial, count);
  hipLaunchKernelGGL(bq_header_kernel, dim3(B), dim3(64), 0, s, partial, radius, hdr);
  hipLaunchKernelGGL(bq_count_kernel, dim3(pblocks), dim3(256), 0, s, xyz, B, N, hdr, cell_of, count);
  hipLaunchKernelGGL(bq_scan_kernel, dim3(B), dim3(1024), 0, s, hdr, count, start, cursor);
  hipLaunchKernelGGL(bq_scatter_kernel, dim3(pblocks), dim3(256), 0, s, xyz, B, N, cell_of, cursor, sorted);
  hipLaunchKernelGGL(bq_query_kernel, dim3((unsigned)(((long long)B * M + 3) / 4)), dim3(256), 0, s, new_xyz, xyz, hdr, start,
                     sorted, idx, B, N, M, radius * radius, nsample);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
