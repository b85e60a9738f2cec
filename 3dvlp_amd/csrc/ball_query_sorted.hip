// Ball query on the spatial sort the pruned FPS has just built for the same cloud — same output, bit for bit, as
// csrc/ball_query.hip (ball_query_gpu.cu:14-59 under the documented fp32 evaluation order), in ONE launch.
//
// SA1 of the backbone samples its centres with vlp3d_furthest_point_sampling_pruned and then queries the same 40 000 points.
// That FPS leaves in its workspace (csrc/fps_cells.h): the points in Hilbert-cell order, the permutation back to the original
// indices, and for each of the 32^3 cells of the scene's bounding box the end of its run.  The grid ball query of
// csrc/ball_query_grid.hip sorted the cloud a second time (bounding box, header, count, scan, scatter: five launches, ~50 MB of
// HBM traffic, before its query kernel); here a wave owns a centre and
//   * turns [c - r', c + r'] into a box of cells (r' = 1.001 r covers the fp32 rounding of the distance test; the cell function
//     is monotone, so every point inside the ball lies in a cell of the box),
//   * lanes fetch the run bounds of "their" cell (the cells of a box are NOT contiguous on the Hilbert curve: one run per cell),
//     a wave scan makes one candidate list of the runs, staged in LDS,
//   * 64 candidates per step are tested with the brute-force kernel's own expression, hits collected by ballot / popcount with
//     their ORIGINAL indices, ranked, and the row — the nsample smallest indices in ascending order, padded with the smallest —
//     is composed in LDS and leaves as whole 4 * nsample-byte stores (the grid kernel's 4-byte rank-scattered stores cost 8.8x
//     the row's bytes in HBM writes, profiles/r03_g_pmc_traffic.txt).
// A box with more than MAX_CAND candidates is walked cell by cell without the list; a ball with more than MAX_HITS points falls
// back to the ordered scan over all points, for that centre only.
#include "common.h"
#include "fps_cells.h"

namespace {

using vlp3d_cells::BBOX_PARTS;
using vlp3d_cells::NCELL;

constexpr int MAX_CAND = 1024;  // candidates of one centre staged in LDS (4 + 4 KB per wave: five workgroups per CU)
constexpr int MAX_HITS = 1024;  // hits of one centre
constexpr int WAVES = 4;

__global__ __launch_bounds__(256) void bq_sorted_kernel(const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all,
                                                        const float4 *__restrict__ pts_all, const int *__restrict__ perm_all,
                                                        const int *__restrict__ hist_all, const float *__restrict__ box_all,
                                                        int *__restrict__ idx_all, int B, int N, int M, float radius,
                                                        float radius2, int nsample) {
  __shared__ int s_cand[WAVES][MAX_CAND];  // sorted positions of the candidates; afterwards the composed output row
  __shared__ int s_hits[WAVES][MAX_HITS];  // original indices of the hits
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // blockIdx % B = scene: consecutive workgroups go to consecutive XCDs, so (for B a multiple of 8) an XCD's L2 holds the
  // sorted points / runs / permutation of its own scenes only (~0.9 MB per 40 000-point scene)
  const int b = blockIdx.x % B;
  const int m = (blockIdx.x / B) * WAVES + wave;
  if (m >= M) return;  // wave-uniform; no workgroup barrier below
  const long long t = (long long)b * M + m;
  const float *c3 = new_xyz_all + t * 3;
  const float cx = c3[0], cy = c3[1], cz = c3[2];
  const float4 *pts = pts_all + (size_t)b * N;
  const int *perm = perm_all + (size_t)b * N;
  const int *hist = hist_all + (size_t)b * NCELL;
  int *cand = s_cand[wave], *hits = s_hits[wave];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  const float *bb = box_all + (size_t)b * 8;  // the scene's bounding box, folded by the pre-pass (wave-uniform: scalar loads)
  const float rm = radius * 1.001f;
  const int x0 = vlp3d_cells::axis_cell(cx - rm, bb[0], bb[3]), x1 = vlp3d_cells::axis_cell(cx + rm, bb[0], bb[3]);
  const int y0 = vlp3d_cells::axis_cell(cy - rm, bb[1], bb[4]), y1 = vlp3d_cells::axis_cell(cy + rm, bb[1], bb[4]);
  const int z0 = vlp3d_cells::axis_cell(cz - rm, bb[2], bb[5]), z1 = vlp3d_cells::axis_cell(cz + rm, bb[2], bb[5]);
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
  const int ncells = nx * ny * nz;
  const float inv_nx = 1.0f / (float)nx, inv_ny = 1.0f / (float)ny;

  // the run (start, length in the sorted order) of cell j of the box, x fastest
  auto run_of = [&](int j, int &rs, int &rl) {
    rs = 0;
    rl = 0;
    if (j < ncells) {
      // j -> (ix, iy, iz); j < 32768 and the divisors <= 32: the float quotient (j + 0.5) / n truncates exactly
      const int r = (int)(((float)j + 0.5f) * inv_nx), ix = j - r * nx;
      const int iz = (int)(((float)r + 0.5f) * inv_ny), iy = r - iz * ny;
      const int h = vlp3d_cells::cell_code((unsigned)(x0 + ix), (unsigned)(y0 + iy), (unsigned)(z0 + iz));
      const int e = hist[h];
      rs = h > 0 ? hist[h - 1] : 0;
      rl = e - rs;
    }
  };
  int *row_out = idx_all + t * nsample;
  int cnt = 0;
  auto test = [&](const float4 &p, int orig, bool live) {  // the scan kernel's expression; hits keep their ORIGINAL index
    const bool hit = live && vlp3d_sumsq3(cx - p.x, cy - p.y, cz - p.z) < radius2;
    const unsigned long long mask = __ballot(hit);
    if (mask != 0ull) {
      const int at = cnt + __popcll(mask & lt_mask);
      if (hit && at < MAX_HITS) hits[at] = orig;
      cnt += __popcll(mask);
    }
  };

  // one run per cell -> one candidate list (sorted positions) in LDS
  int total = 0;
  for (int j0 = 0; j0 < ncells && total <= MAX_CAND; j0 += 64) {
    int rs, rl;
    run_of(j0 + lane, rs, rl);
    int inc = rl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    const int at = total + inc - rl;
    for (int k = 0; __ballot(k < rl) != 0ull; ++k)
      if (k < rl && at + k < MAX_CAND) cand[at + k] = rs + k;
    total += __shfl(inc, 63);
  }
  if (total <= MAX_CAND) {
    for (int c0 = 0; c0 < total; c0 += 256) {  // four chunks of 64 candidates requested together
      float4 p[4];
      int orig[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 64 * u + lane;
        const int pos = cand[c < total ? c : total - 1];
        p[u] = pts[pos];
        orig[u] = perm[pos];  // requested with the point (only a hit needs it, but a dependent load later costs a round trip)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) test(p[u], orig[u], c0 + 64 * u + lane < total);
    }
  } else {
    // a box too crowded for the LDS list (3 of SA1's 16 384 centres on the bench scenes): the same cells again, a lane walking
    // its own cell's run — uneven runs leave lanes idle, but nothing is limited by a capacity and no point outside the box is
    // touched (the brute-force scan of all N points here made three waves the kernel's whole duration: 28 -> 140 us)
    for (int j0 = 0; j0 < ncells; j0 += 64) {
      int rs, rl;
      run_of(j0 + lane, rs, rl);
      for (int k = 0; __ballot(k < rl) != 0ull; ++k) {
        const int pos = rs + (k < rl ? k : 0);
        test(pts[pos < N ? pos : N - 1], perm[pos < N ? pos : N - 1], k < rl);
      }
    }
  }
  if (cnt > MAX_HITS || nsample > MAX_CAND) {  // more than 1024 points inside ONE ball (or a row that does not fit the LDS
    // staging): the ordered scan over all points, for this centre only
    const float *xyz = xyz_all + (size_t)b * N * 3;
    int n = 0, first = 0;
    for (int k0 = 0; k0 < N && n < nsample; k0 += 64) {
      const int k = k0 + lane;
      const bool hit = k < N && vlp3d_sumsq3(cx - xyz[k * 3], cy - xyz[k * 3 + 1], cz - xyz[k * 3 + 2]) < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull) {
        if (n == 0) first = k0 + __ffsll((long long)mask) - 1;
        const int at = n + __popcll(mask & lt_mask);
        if (hit && at < nsample) row_out[at] = k;
        n += __popcll(mask);
      }
    }
    n = n < nsample ? n : nsample;
    for (int l = n + lane; l < nsample; l += 64) row_out[l] = first;
    return;
  }
  // rank every hit (number of hits with a smaller original index; indices are distinct) and compose the row: the nsample
  // smallest in ascending order, then the smallest again (empty ball: zeros, like the reference's pre-zeroed output).
  // Ranks by LDS BROADCAST reads, sixteen hits per trip (four ds_read_b128 in flight): stage timing of the first form — one
  // cross-lane read per hit, each waited for — showed 125 of the kernel's 150 us here.
  int *row = cand;  // the row is composed in LDS, over the candidate list (no longer needed)
  {
    const int cnt16 = (cnt + 15) & ~15;
    for (int l = cnt + lane; l < cnt16; l += 64) hits[l] = 0x7fffffff;  // padding never counts as "smaller"
  }
  const int nj = (cnt + 63) >> 6;  // wave-uniform, <= MAX_HITS / 64
  if (nj <= 1) {  // the common case: one hit per lane
    const int me = lane < cnt ? hits[lane] : 0x7fffffff;
    int rk = 0;
    for (int i = 0; i < cnt; i += 16) {
      const int4 a = *reinterpret_cast<const int4 *>(hits + i), b4 = *reinterpret_cast<const int4 *>(hits + i + 4);
      const int4 c4 = *reinterpret_cast<const int4 *>(hits + i + 8), d = *reinterpret_cast<const int4 *>(hits + i + 12);
      rk += (a.x < me) + (a.y < me) + (a.z < me) + (a.w < me) + (b4.x < me) + (b4.y < me) + (b4.z < me) + (b4.w < me) +
            (c4.x < me) + (c4.y < me) + (c4.z < me) + (c4.w < me) + (d.x < me) + (d.y < me) + (d.z < me) + (d.w < me);
    }
    if (lane < cnt && rk < nsample) row[rk] = me;
  } else {
    int mine[MAX_HITS / 64], rank[MAX_HITS / 64];
#pragma unroll
    for (int j = 0; j < MAX_HITS / 64; ++j) {
      mine[j] = (j * 64 + lane < cnt) ? hits[j * 64 + lane] : 0x7fffffff;
      rank[j] = 0;
    }
    for (int i = 0; i < cnt; i += 4) {
      const int4 v = *reinterpret_cast<const int4 *>(hits + i);  // LDS broadcast
#pragma unroll
      for (int j = 0; j < MAX_HITS / 64; ++j)
        if (j < nj) rank[j] += (v.x < mine[j]) + (v.y < mine[j]) + (v.z < mine[j]) + (v.w < mine[j]);
    }
#pragma unroll
    for (int j = 0; j < MAX_HITS / 64; ++j)
      if (j < nj && j * 64 + lane < cnt && rank[j] < nsample) row[rank[j]] = mine[j];
  }
  const int n = cnt < nsample ? cnt : nsample;
  const int pad = cnt > 0 ? row[0] : 0;  // the hit of rank 0 (LDS broadcast; the wave's ds operations complete in order)
  for (int l = n + lane; l < nsample; l += 64) row[l] = pad;
  for (int l = lane; l < nsample; l += 64) row_out[l] = row[l];
}

}  // namespace

// Ball query of `new_xyz` against the cloud whose pruned FPS has just filled `fps_workspace`
// (vlp3d_furthest_point_sampling_pruned(xyz, B, N, ., fps_workspace, ...) on the SAME xyz, same B and N, not overwritten since).
// idx (B, M, nsample): identical to vlp3d_ball_query.
extern "C" int vlp3d_ball_query_sorted(const float *new_xyz, const float *xyz, int B, int N, int M, float radius, int nsample,
                                       const void *fps_workspace, long long fps_workspace_bytes, int *idx, void *stream) {
  if (!new_xyz || !xyz || !idx || !fps_workspace || B < 1 || N < 1 || M < 1 || nsample < 1 || !(radius > 0.f)) return VLP3D_EINVAL;
  if ((long long)N * 3 >= (1ll << 31) || (long long)B * N >= (1ll << 31) || (long long)B * M >= (1ll << 31)) return VLP3D_EINVAL;
  if (fps_workspace_bytes < vlp3d_cells::workspace_bytes(B, N)) return VLP3D_EINVAL;
  const vlp3d_cells::Workspace ws = vlp3d_cells::workspace_layout(const_cast<void *>(fps_workspace), B, N);
  const long long blocks = (long long)B * ((M + WAVES - 1) / WAVES);
  if (blocks >= (1ll << 31)) return VLP3D_EINVAL;
  hipLaunchKernelGGL(bq_sorted_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, new_xyz, xyz, ws.pts, ws.perm,
                     ws.hist, ws.box, idx, B, N, M, radius, radius * radius, nsample);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
