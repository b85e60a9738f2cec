// The spatial sort that the pruned FPS builds in its pre-pass (csrc/fps_pruned.hip) and that the sorted ball query
// (csrc/ball_query_sorted.hip) reads again: 32 x 32 x 32 cells over the scene's bounding box, ordered along the 3-D Hilbert curve.
#pragma once
#include "common.h"

namespace vlp3d_cells {

constexpr int CELL_BITS = 5;                  // per axis
constexpr int NCELL = 1 << (3 * CELL_BITS);   // 32768 cells per scene
constexpr int BBOX_PARTS = 16;                // partial bounding boxes per scene (fps_bbox_kernel)

__host__ __device__ __forceinline__ size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Layout of the workspace of vlp3d_furthest_point_sampling_pruned (vlp3d_fps_workspace_bytes(B, N) bytes):
//   pts    (B, N) float4   points in cell order: x, y, z, initial running minimum (1e10, or -1 inside the FPS skip ball)
//   perm   (B, N) int      sorted position -> original index
//   cellid (B, N) int      cell code of every original point
//   hist   (B, NCELL) int  after the pre-pass: END offset of every cell's run in the sorted order (start = end of the cell before)
//   bbox   (B, BBOX_PARTS, 6) float  partial boxes: min x y z, max x y z
//   box    (B, 8) float    the scene's bounding box (the partial boxes folded: min x y z, max x y z, 0, 0)
struct Workspace {
  float4 *pts;
  int *perm, *cellid, *hist;
  float *bbox, *box;
};
__host__ __device__ inline Workspace workspace_layout(void *base, int B, int N) {
  char *w = (char *)base;
  Workspace s;
  s.pts = (float4 *)w;
  w += align256((size_t)B * N * 16);
  s.perm = (int *)w;
  w += align256((size_t)B * N * 4);
  s.cellid = (int *)w;
  w += align256((size_t)B * N * 4);
  s.hist = (int *)w;
  w += align256((size_t)B * NCELL * 4);
  s.bbox = (float *)w;
  w += align256((size_t)B * BBOX_PARTS * 6 * 4);
  s.box = (float *)w;
  return s;
}
__host__ __device__ inline long long workspace_bytes(int B, int N) {
  return (long long)(align256((size_t)B * N * 16) + align256((size_t)B * N * 4) * 2 + align256((size_t)B * NCELL * 4) +
                     align256((size_t)B * BBOX_PARTS * 6 * 4) + align256((size_t)B * 8 * 4));
}

// cell coordinate of p on an axis whose points span [lo, hi]: monotone non-decreasing in p (subtraction, division by a
// positive number, multiplication by 32 and the truncation all are), 0 for NaN
__device__ __forceinline__ int axis_cell(float p, float lo, float hi) {
  const float ext = hi - lo;
  const float t = ext > 0.f ? (p - lo) / ext * 32.f : 0.f;
  int c = (int)t;
  return c < 0 ? 0 : (c > 31 ? 31 : c);
}

__device__ __forceinline__ unsigned spread5(unsigned v) {  // 5 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}

// position of cell (q0, q1, q2) on the 3-D Hilbert curve (Skilling's transpose form, 5 bits per axis): consecutive cells are
// face neighbours, so a slot (64 consecutive sorted points) never straddles one of the Z-order curve's long jumps and its
// bounding box is tighter.  -DVLP3D_FPS_MORTON: the Z-order code instead (experiments).
__device__ __forceinline__ int cell_code(unsigned q0, unsigned q1, unsigned q2) {
#ifdef VLP3D_FPS_MORTON
  return (int)(spread5(q0) | (spread5(q1) << 1) | (spread5(q2) << 2));
#else
  unsigned q[3] = {q0, q1, q2};
  const unsigned M = 1u << (CELL_BITS - 1);
  for (unsigned Q = M; Q > 1; Q >>= 1) {
    const unsigned P = Q - 1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (q[a] & Q) {
        q[0] ^= P;
      } else {
        const unsigned t = (q[0] ^ q[a]) & P;
        q[0] ^= t;
        q[a] ^= t;
      }
    }
  }
  q[1] ^= q[0];
  q[2] ^= q[1];
  unsigned t = 0;
  for (unsigned Q = M; Q > 1; Q >>= 1)
    if (q[2] & Q) t ^= Q - 1;
  q[0] ^= t; q[1] ^= t; q[2] ^= t;
  return (int)((spread5(q[0]) << 2) | (spread5(q[1]) << 1) | spread5(q[2]));
#endif
}

}  // namespace vlp3d_cells
