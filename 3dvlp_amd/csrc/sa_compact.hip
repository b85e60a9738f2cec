// Compact row map of a grouped MLP (csrc/sa_mlp.hip, bf16 configuration).
//
// ball_query pads a ball that holds fewer than nsample points with copies of its FIRST neighbour
// (ball_query_gpu.cu:14-49), and the reference pushes the padded (B, C, npoint, nsample) tensor through the whole
// SharedMLP.  Every copy of a row yields the same pre-activations; BatchNorm sees the copies only through their
// multiplicity in the batch sums and max-pooling not at all.  The stack is therefore evaluated on the DISTINCT rows:
//   * batch sums  sum_r f(y_r)             = sum_u w_u f(y_u)      (w_u = multiplicity: 1, or nsample - cnt + 1 for the
//                                                                 ball's first neighbour)
//   * BatchNorm backward of one copy        dy = k1 (g - k2 - yhat k3); summed over the copies of u:
//                                           dY_u = k1 (G_u - w_u (k2 + yhat_u k3)),  G_u = sum of the copies' g
//   * everything downstream of dY (dX = dY W^T, dW = dY^T A, ReLU masks, the scatter) is linear in it.
// At cfg2 39 % of SA1's and 18 % of SA2's rows are distinct (tools/dup_fraction.py).
//
// The map is computed once per batch next to the ball query (side stream): rowptr[bm] = first compact row of ball bm
// (rowptr[B*M] = number of compact rows P), crow[r'] = (global point row b*N + idx, (bm << 8) | s, float bits of w, 0);
// rows P .. roundup32(P)-1 are dummies (point 0, s = 255, w = 0) so that the kernels can work on whole 32-row tiles.
#include <stdlib.h>

#include "common.h"

namespace {

// cnt[bm] = number of distinct neighbours = position of the first repeat of idx[bm][0] (S if there is none)
__global__ __launch_bounds__(256) void sa_cnt_kernel(const int *__restrict__ idx, int nb, int S, int *__restrict__ cnt) {
  const int bm = blockIdx.x * 256 + threadIdx.x;
  if (bm >= nb) return;
  const int *p = idx + (long long)bm * S;
  const int first = p[0];
  int c = S;
  for (int s = 1; s < S; ++s)
    if (p[s] == first) { c = s; break; }
  cnt[bm] = c;
}

// in-place exclusive scan of v[0..n) by ONE workgroup; v[n] = total.  Four consecutive entries per thread and round (16 384
// balls: 4 rounds instead of 16), shuffle scan inside a wave, every wave scans the 16 wave totals.
__global__ __launch_bounds__(1024) void sa_scan_kernel(int *__restrict__ v, int n) {
  __shared__ int wtot[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 4096) {
    const int i = c0 + 4 * threadIdx.x;
    int a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = i + k < n ? v[i + k] : 0;
    const int mine = (a[0] + a[1]) + (a[2] + a[3]);
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int w = lane < 16 ? wtot[lane] : 0;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(w, off);
      if (lane >= off) w += t;
    }
    const int before = wave > 0 ? __shfl(w, wave - 1) : 0, all = __shfl(w, 15);
    int run = carry_s + before + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < n) v[i + k] = run;
      run += a[k];
    }
    __syncthreads();  // every wave has read carry_s and wtot
    if (threadIdx.x == 0) carry_s += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) v[n] = carry_s;
}

__global__ __launch_bounds__(256) void sa_fill_kernel(const int *__restrict__ idx, int B, int N, int M, int S,
                                                      const int *__restrict__ rowptr, int4 *__restrict__ crow) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long R = (long long)B * M * S;
  const int nb = B * M;
  if (t < 32) {  // dummy rows that complete the last tile
    const int P = rowptr[nb];
    const int r = P + (int)t;
    if (r < ((P + 31) & ~31)) crow[r] = make_int4(0, 255, __float_as_int(0.f), 0);
  }
  if (t >= R) return;
  const int bm = (int)(t / S), s = (int)(t - (long long)bm * S);
  const int r0 = rowptr[bm], cnt = rowptr[bm + 1] - r0;
  if (s >= cnt) return;
  const int b = bm / M;
  const float w = s == 0 ? (float)(S - cnt + 1) : 1.f;
  crow[r0 + s] = make_int4(b * N + idx[t], (bm << 8) | s, __float_as_int(w), 0);
}

// ---- inverse map: for every source point the rows that gather it (CSR) ---------------------------------------------------
// The backward of a gather layer adds each row's input gradient to its source point.  With this map the sum is a GATHER
// per point (no atomics, csrc/sa_gather_sum.hip).  rows = compact rows u < P (crow given) or all B*M*S padded rows.
__global__ __launch_bounds__(256) void sa_inv_count_kernel(const int *__restrict__ idx, const int4 *__restrict__ crow,
                                                           const int *__restrict__ rowptr, int nb, int N, long long MS,
                                                           long long R, int *__restrict__ cnt) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = crow ? (long long)rowptr[nb] : R;
  if (t >= n) return;
  const int p = crow ? crow[t].x : (int)((t / MS) * N) + idx[t];
  atomicAdd(cnt + p, 1);
}
__global__ __launch_bounds__(256) void sa_inv_fill_kernel(const int *__restrict__ idx, const int4 *__restrict__ crow,
                                                          const int *__restrict__ rowptr, int nb, int N, long long MS,
                                                          long long R, const int *__restrict__ start, int *__restrict__ cursor,
                                                          int *__restrict__ rows) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = crow ? (long long)rowptr[nb] : R;
  if (t >= n) return;
  const int p = crow ? crow[t].x : (int)((t / MS) * N) + idx[t];
  rows[start[p] + atomicAdd(cursor + p, 1)] = (int)t;
}


// The whole inverse map of ONE cloud in one workgroup (N <= SA_INV_BLOCK_N points): its rows are a contiguous range of the
// (compact) row list, so counters, the exclusive scan and the fill cursors stay in LDS and the cloud's first slot is the
// range's start — no cross-workgroup scan.  One launch instead of five (two clears, count, scan, fill: ~5 us each on the
// geometry stream, five inverse maps per step).
constexpr int SA_INV_BLOCK_N = 8192;
__global__ __launch_bounds__(1024) void sa_inverse_block_kernel(const int *__restrict__ idx, const int4 *__restrict__ crow,
                                                                const int *__restrict__ rowptr, int N, int M, long long MS,
                                                                int *__restrict__ inv_start, int *__restrict__ inv_rows) {
  extern __shared__ int sm_inv[];
  int *cnt = sm_inv, *pos = sm_inv + N;
  __shared__ int wtot[16];
  __shared__ int carry_s;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = crow ? (long long)rowptr[b * M] : b * MS, t1 = crow ? (long long)rowptr[(b + 1) * M] : (b + 1) * MS;
  const int p0 = b * N;
  for (int i = threadIdx.x; i < N; i += 1024) cnt[i] = 0;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (long long t = t0 + threadIdx.x; t < t1; t += 1024) atomicAdd(cnt + ((crow ? crow[t].x : p0 + idx[t]) - p0), 1);
  __syncthreads();
  for (int c0 = 0; c0 < N; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int mine = i < N ? cnt[i] : 0;
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int w = lane < 16 ? wtot[lane] : 0;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(w, off);
      if (lane >= off) w += t;
    }
    const int before = wave > 0 ? __shfl(w, wave - 1) : 0, all = __shfl(w, 15);
    if (i < N) {
      const int ex = carry_s + before + inc - mine;
      pos[i] = ex;
      inv_start[p0 + i] = (int)t0 + ex;
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += all;
    __syncthreads();
  }
  if (b == (int)gridDim.x - 1 && threadIdx.x == 0) inv_start[p0 + N] = (int)t1;
  for (long long t = t0 + threadIdx.x; t < t1; t += 1024)
    inv_rows[t0 + atomicAdd(pos + ((crow ? crow[t].x : p0 + idx[t]) - p0), 1)] = (int)t;
}

}  // namespace

// inv_start (B*N + 1), inv_rows (B*M*S), cursor (B*N) scratch.  crow / rowptr: the compact map of vlp3d_sa_compact or NULL.
// The order of a point's rows follows the atomics (any order: the consumer sums them).
extern "C" int vlp3d_sa_inverse(const int *idx, const void *crow, const int *rowptr, int B, int N, int M, int S, int *inv_start,
                                int *inv_rows, int *cursor, void *stream) {
  if (!idx || !inv_start || !inv_rows || !cursor || (crow && !rowptr) || B < 1 || N < 1 || M < 1 || S < 1 ||
      (long long)B * N >= (1ll << 30) || (long long)B * M * S >= (1ll << 31))
    return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int np = B * N, nb = B * M;
  const long long R = (long long)nb * S, MS = (long long)M * S;
  static const bool one_launch = !(getenv("VLP3D_SA_INVERSE_BLOCK") && atoi(getenv("VLP3D_SA_INVERSE_BLOCK")) == 0);
  if (one_launch && N <= SA_INV_BLOCK_N) {
    hipLaunchKernelGGL(sa_inverse_block_kernel, dim3(B), dim3(1024), 2 * N * sizeof(int), s, idx, (const int4 *)crow, rowptr, N, M,
                       MS, inv_start, inv_rows);
    VLP3D_LAUNCH_CHECK();
    return VLP3D_OK;
  }
  hipError_t e = vlp3d_zero_words(inv_start, (size_t)np + 1, s);
  if (e == hipSuccess) e = vlp3d_zero_words(cursor, (size_t)np, s);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((unsigned)((R + 255) / 256));
  hipLaunchKernelGGL(sa_inv_count_kernel, grid, dim3(256), 0, s, idx, (const int4 *)crow, rowptr, nb, N, MS, R, inv_start);
  hipLaunchKernelGGL(sa_scan_kernel, dim3(1), dim3(1024), 0, s, inv_start, np);
  hipLaunchKernelGGL(sa_inv_fill_kernel, grid, dim3(256), 0, s, idx, (const int4 *)crow, rowptr, nb, N, MS, R, inv_start, cursor,
                     inv_rows);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// rowptr: (B*M + 1) ints; crow: (B*M*S) int4 (worst case: every row distinct).  S <= 255, B*M < 2^23, B*N < 2^31.
extern "C" int vlp3d_sa_compact(const int *idx, int B, int N, int M, int S, int *rowptr, void *crow, void *stream) {
  if (!idx || !rowptr || !crow || B < 1 || N < 1 || M < 1 || S < 1 || S > 255 || (long long)B * M >= (1ll << 23) ||
      (long long)B * N >= (1ll << 31) || (long long)B * M * S >= (1ll << 31))
    return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nb = B * M;
  hipLaunchKernelGGL(sa_cnt_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, idx, nb, S, rowptr);
  hipLaunchKernelGGL(sa_scan_kernel, dim3(1), dim3(1024), 0, s, rowptr, nb);
  const long long R = (long long)nb * S;
  hipLaunchKernelGGL(sa_fill_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, idx, B, N, M, S, rowptr,
                     (int4 *)crow);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
