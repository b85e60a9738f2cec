// Training-time scene augmentation on the device — what the reference's loader does per scene in numpy on
// DataLoader workers (lib/joint/dataset.py:653-690 with utils/utils_fn.py:28-142 flip / rotate / scale / translate and
// data/scannet/model_util_scannet.py:48-80 for the boxes), with the votes recomputed AFTER augmentation from the instance
// labels.  The random draws stay on the host (11 numbers + the composed rotation per scene: vlp3d_augment_param_floats()
// floats, layout of oracle/augment.py: draw_params); everything that touches 40 000 points runs here, on the copy stream:
//   augment_points   x' = ((flip x) M) * s + t per point, height channel * s_z; the per-instance POINT bounding boxes of the
//                    augmented cloud are reduced on the way (LDS table per workgroup, ordered-int atomics), one launch
//   augment_votes    vote = 0.5 (min + max) of the point's instance - x, three copies; mask = instance annotated
//   augment_boxes    the M axis-aligned GT boxes through the same flips, the three axis rotations (enclosing aligned box,
//                    model_util_scannet.py:59-71), scale and translation
#include "common.h"

namespace {

constexpr int PF = 24;    // floats per scene
constexpr int IMAX = 256;  // instance ids per scene (ScanNet scenes hold < 200)

__device__ __forceinline__ int fkey(float f) {  // order-preserving float -> int
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float funkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void augment_points_kernel(float *__restrict__ pc, int N, int C, int height_col,
                                                             const float *__restrict__ params, const int *__restrict__ inst,
                                                             int I, int *__restrict__ ibox) {
  __shared__ int tab[IMAX * 6];
  const int b = blockIdx.y;
  const float *p = params + (size_t)b * PF;
  for (int i = threadIdx.x; i < I * 6; i += 256) tab[i] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    float *row = pc + ((size_t)b * N + n) * C;
    float x = row[0], y = row[1], z = row[2];
    if (p[0] != 0.f) x = -x;
    if (p[1] != 0.f) y = -y;
    // row vector times M (utils_fn.py:101-104), evaluated left to right like np.dot
    const float rx = x * p[12] + y * p[15] + z * p[18];
    const float ry = x * p[13] + y * p[16] + z * p[19];
    const float rz = x * p[14] + y * p[17] + z * p[20];
    x = rx * p[5] + p[8];
    y = ry * p[6] + p[9];
    z = rz * p[7] + p[10];
    row[0] = x; row[1] = y; row[2] = z;
    if (height_col >= 0) row[height_col] *= p[7];
    if (inst) {
      const int id = inst[(size_t)b * N + n];
      if (id >= 0 && id < I) {
        atomicMin(&tab[id * 6 + 0], fkey(x)); atomicMin(&tab[id * 6 + 1], fkey(y)); atomicMin(&tab[id * 6 + 2], fkey(z));
        atomicMax(&tab[id * 6 + 3], fkey(x)); atomicMax(&tab[id * 6 + 4], fkey(y)); atomicMax(&tab[id * 6 + 5], fkey(z));
      }
    }
  }
  if (!inst) return;
  __syncthreads();
  for (int i = threadIdx.x; i < I * 6; i += 256) {
    const int v = tab[i];
    if ((i % 6) < 3) { if (v != 0x7fffffff) atomicMin(&ibox[(size_t)b * I * 6 + i], v); }
    else if (v != (int)0x80000000) atomicMax(&ibox[(size_t)b * I * 6 + i], v);
  }
}

__global__ __launch_bounds__(256) void augment_ibox_init_kernel(int *__restrict__ ibox, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) ibox[i] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
}

__global__ __launch_bounds__(256) void augment_votes_kernel(const float *__restrict__ pc, int N, int C, const int *__restrict__ inst,
                                                            int I, const int *__restrict__ ibox,
                                                            const unsigned char *__restrict__ valid,
                                                            float *__restrict__ vote, float *__restrict__ mask_f,
                                                            long long *__restrict__ mask_i) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const size_t pn = (size_t)b * N + n;
  const int id = inst[pn];
  float v[3] = {0.f, 0.f, 0.f};
  float m = 0.f;
  if (id >= 0 && id < I && valid[(size_t)b * I + id]) {
    const int *bx = ibox + ((size_t)b * I + id) * 6;
    const float *row = pc + pn * C;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = 0.5f * (funkey(bx[c]) + funkey(bx[3 + c])) - row[c];
    m = 1.f;
  }
  float *o = vote + pn * 9;
#pragma unroll
  for (int r = 0; r < 3; ++r) { o[3 * r] = v[0]; o[3 * r + 1] = v[1]; o[3 * r + 2] = v[2]; }
  if (mask_f) mask_f[pn] = m;
  if (mask_i) mask_i[pn] = (long long)m;
}

// one axis rotation of an aligned box: centre times R^T, the two in-plane lengths from the rotated half-extent corners
__device__ __forceinline__ void rot_box(float (&c)[3], float (&l)[3], float ang, int axis) {
  const float cs = cosf(ang), sn = sinf(ang);
  float R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (axis == 0) { R[1][1] = cs; R[1][2] = -sn; R[2][1] = sn; R[2][2] = cs; }
  else if (axis == 1) { R[0][0] = cs; R[0][2] = sn; R[2][0] = -sn; R[2][2] = cs; }
  else { R[0][0] = cs; R[0][1] = -sn; R[1][0] = sn; R[1][1] = cs; }
  const float nc[3] = {c[0] * R[0][0] + c[1] * R[0][1] + c[2] * R[0][2], c[0] * R[1][0] + c[1] * R[1][1] + c[2] * R[1][2],
                       c[0] * R[2][0] + c[1] * R[2][1] + c[2] * R[2][2]};
  const int i1 = axis == 0 ? 1 : 0, i2 = axis == 2 ? 1 : 2;
  const float d1 = l[i1] * 0.5f, d2 = l[i2] * 0.5f;
  float m1 = -INFINITY, m2 = -INFINITY;
  const float sg[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // crnrs = (s1 d1, s2 d2, 0) R^T: columns 0 and 1 (model_util_scannet.py:62-68)
    const float a = sg[k][0] * d1, bq = sg[k][1] * d2;
    m1 = fmaxf(m1, a * R[0][0] + bq * R[0][1]);
    m2 = fmaxf(m2, a * R[1][0] + bq * R[1][1]);
  }
  c[0] = nc[0]; c[1] = nc[1]; c[2] = nc[2];
  l[i1] = 2.f * m1;
  l[i2] = 2.f * m2;
}

__global__ __launch_bounds__(256) void augment_boxes_kernel(const float *__restrict__ boxes, int M,
                                                            const float *__restrict__ params, float *__restrict__ out, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float *p = params + (size_t)(i / M) * PF;
  const float *bx = boxes + i * 6;
  float c[3] = {bx[0], bx[1], bx[2]}, l[3] = {bx[3], bx[4], bx[5]};
  if (p[0] != 0.f) c[0] = -c[0];
  if (p[1] != 0.f) c[1] = -c[1];
  rot_box(c, l, p[2], 0);
  rot_box(c, l, p[3], 1);
  rot_box(c, l, p[4], 2);
  float *o = out + i * 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o[k] = c[k] * p[5 + k] + p[8 + k];
    o[3 + k] = l[k] * p[5 + k];
  }
}

}  // namespace

extern "C" int vlp3d_augment_param_floats(void) { return PF; }
extern "C" int vlp3d_augment_max_instances(void) { return IMAX; }

// pc (B,N,C) augmented IN PLACE (columns 0..2, and height_col when >= 0).  inst (B,N) int32 + ibox (B,I,6) int32 scratch:
// also reduce the per-instance point bounding boxes (NULL: skip).  params (B, vlp3d_augment_param_floats()) on the device.
extern "C" int vlp3d_augment_points(float *pc, int B, int N, int C, int height_col, const float *params, const int *inst, int I,
                                    int *ibox, void *stream) {
  if (!pc || !params || B < 1 || N < 1 || C < 3 || height_col >= C || (inst && (!ibox || I < 1 || I > IMAX))) return -22;
  hipStream_t s = (hipStream_t)stream;
  if (inst) {
    const long long n = (long long)B * I * 6;
    hipLaunchKernelGGL(augment_ibox_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ibox, n);
    VLP3D_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(augment_points_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, pc, N, C, height_col, params, inst, I, ibox);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

// vote (B,N,9), mask_f (B,N) float and / or mask_i (B,N) int64 from the augmented cloud and the boxes of vlp3d_augment_points.
extern "C" int vlp3d_augment_votes(const float *pc, int B, int N, int C, const int *inst, int I, const int *ibox,
                                   const unsigned char *valid, float *vote, float *mask_f, long long *mask_i, void *stream) {
  if (!pc || !inst || !ibox || !valid || !vote || B < 1 || N < 1 || C < 3 || I < 1 || I > IMAX) return -22;
  hipLaunchKernelGGL(augment_votes_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, pc, N, C, inst, I, ibox,
                     valid, vote, mask_f, mask_i);
  VLP3D_LAUNCH_CHECK();
  return 0;
}

// boxes (B,M,6) [centre | lengths] -> out (B,M,6)
extern "C" int vlp3d_augment_boxes(const float *boxes, int B, int M, const float *params, float *out, void *stream) {
  if (!boxes || !params || !out || B < 1 || M < 1) return -22;
  const long long total = (long long)B * M;
  hipLaunchKernelGGL(augment_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes, M,
                     params, out, total);
  VLP3D_LAUNCH_CHECK();
  return 0;
}
