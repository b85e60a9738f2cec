// Box decode of the proposal module for gfx950 — replaces the ~30 element-wise launches (and as many in the
// autograd backward) of models/proposal_module/proposal_module_fcos.py:94-144 (decode_pred_box) plus
// utils/box_util.py:361-385 (get_3d_box_batch, rotation roty_batch :324-338) by one kernel each way.
// One thread per proposal:
//   heading_class = argmax(heading_scores)                         (first maximum, like torch.argmax)
//   heading       = class * (2 pi / NH) + heading_residuals[class]
//   size          = rois[0:3] + rois[3:6];   half = (rois[0:3] - rois[3:6]) / 2
//   centre        = vote_xyz - [hx*c + hy*s, -hx*s + hy*c, hz]     (c, s = cos/sin(heading))
//   corners[j]    = rot_y(heading) * (sign_j * size / 2) + centre   (detached in the reference: no gradient)
// Backward returns the gradients w.r.t. rois, heading_residuals and vote_xyz from those of heading/size/centre.
#include "common.h"

namespace {

__constant__ float kSign[8][3] = {{1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, 1},
                                  {1, 1, -1}, {1, -1, -1}, {-1, -1, -1}, {-1, 1, -1}};

__device__ __forceinline__ int argmax_first(const float *__restrict__ v, int n) {
  int best = 0;
  float bv = v[0];
  for (int i = 1; i < n; ++i) {
    const float x = v[i];
    if (x > bv || (x != x && bv == bv)) {  // NaN counts as the maximum, as in torch.argmax
      bv = x;
      best = i;
    }
  }
  return best;
}

__global__ __launch_bounds__(256) void box_decode_fwd_kernel(const float *__restrict__ vote_xyz,
                                                             const float *__restrict__ heading_scores,
                                                             const float *__restrict__ heading_residuals,
                                                             const float *__restrict__ rois, int n, int NH,
                                                             float *__restrict__ heading, float *__restrict__ size,
                                                             float *__restrict__ centre, float *__restrict__ corners,
                                                             int *__restrict__ heading_class) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int cls = argmax_first(heading_scores + (long long)p * NH, NH);
  const float per_class = (float)(2.0 * 3.14159265358979323846 / (double)NH);
  const float hd = (float)cls * per_class + heading_residuals[(long long)p * NH + cls];
  const float *r = rois + (long long)p * 6;
  const float sx = r[0] + r[3], sy = r[1] + r[4], sz = r[2] + r[5];
  const float hx = (r[0] - r[3]) / 2, hy = (r[1] - r[4]) / 2, hz = (r[2] - r[5]) / 2;
  const float c = cosf(hd), s = sinf(hd);
  const float *v = vote_xyz + (long long)p * 3;
  const float cx = v[0] - (hx * c + hy * s), cy = v[1] - (-hx * s + hy * c), cz = v[2] - hz;
  heading[p] = hd;
  heading_class[p] = cls;
  size[3 * p + 0] = sx; size[3 * p + 1] = sy; size[3 * p + 2] = sz;
  centre[3 * p + 0] = cx; centre[3 * p + 1] = cy; centre[3 * p + 2] = cz;
  float *o = corners + (long long)p * 24;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = kSign[j][0] * (sx * 0.5f), y = kSign[j][1] * (sy * 0.5f), z = kSign[j][2] * (sz * 0.5f);
    o[3 * j + 0] = (c * x + s * z) + cx;
    o[3 * j + 1] = y + cy;
    o[3 * j + 2] = (-s * x + c * z) + cz;
  }
}

__global__ __launch_bounds__(256) void box_decode_bwd_kernel(const float *__restrict__ rois,
                                                             const float *__restrict__ heading,
                                                             const int *__restrict__ heading_class,
                                                             const float *__restrict__ d_heading,
                                                             const float *__restrict__ d_size,
                                                             const float *__restrict__ d_centre, int n, int NH,
                                                             float *__restrict__ d_rois,
                                                             float *__restrict__ d_residuals,
                                                             float *__restrict__ d_vote_xyz) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float *r = rois + (long long)p * 6;
  const float hx = (r[0] - r[3]) / 2, hy = (r[1] - r[4]) / 2;
  const float hd = heading[p];
  const float c = cosf(hd), s = sinf(hd);
  float gsx = 0.f, gsy = 0.f, gsz = 0.f, gcx = 0.f, gcy = 0.f, gcz = 0.f;
  if (d_size) { gsx = d_size[3 * p]; gsy = d_size[3 * p + 1]; gsz = d_size[3 * p + 2]; }
  if (d_centre) { gcx = d_centre[3 * p]; gcy = d_centre[3 * p + 1]; gcz = d_centre[3 * p + 2]; }
  // centre = vote - off, off = [hx*c + hy*s, -hx*s + hy*c, hz]
  const float gox = -gcx, goy = -gcy, goz = -gcz;
  const float ghx = gox * c - goy * s, ghy = gox * s + goy * c, ghz = goz;
  float gh = gox * (-hx * s + hy * c) + goy * (-hx * c - hy * s);
  if (d_heading) gh += d_heading[p];
  float *o = d_rois + (long long)p * 6;
  o[0] = gsx + 0.5f * ghx; o[1] = gsy + 0.5f * ghy; o[2] = gsz + 0.5f * ghz;
  o[3] = gsx - 0.5f * ghx; o[4] = gsy - 0.5f * ghy; o[5] = gsz - 0.5f * ghz;
  const int cls = heading_class[p];
  for (int i = 0; i < NH; ++i) d_residuals[(long long)p * NH + i] = i == cls ? gh : 0.f;
  d_vote_xyz[3 * p + 0] = gcx; d_vote_xyz[3 * p + 1] = gcy; d_vote_xyz[3 * p + 2] = gcz;
}

}  // namespace

extern "C" int vlp3d_box_decode_fwd(const float *vote_xyz, const float *heading_scores, const float *heading_residuals,
                                    const float *rois, int n, int NH, float *heading, float *size, float *centre,
                                    float *corners, int *heading_class, void *stream) {
  if (!vote_xyz || !heading_scores || !heading_residuals || !rois || !heading || !size || !centre || !corners ||
      !heading_class || n < 1 || NH < 1)
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(box_decode_fwd_kernel, dim3(vlp3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, vote_xyz,
                     heading_scores, heading_residuals, rois, n, NH, heading, size, centre, corners, heading_class);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

extern "C" int vlp3d_box_decode_bwd(const float *rois, const float *heading, const int *heading_class,
                                    const float *d_heading, const float *d_size, const float *d_centre, int n, int NH,
                                    float *d_rois, float *d_residuals, float *d_vote_xyz, void *stream) {
  if (!rois || !heading || !heading_class || !d_rois || !d_residuals || !d_vote_xyz || n < 1 || NH < 1)
    return VLP3D_EINVAL;
  hipLaunchKernelGGL(box_decode_bwd_kernel, dim3(vlp3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, heading,
                     heading_class, d_heading, d_size, d_centre, n, NH, d_rois, d_residuals, d_vote_xyz);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
