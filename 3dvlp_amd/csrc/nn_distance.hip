// nn_distance for gfx950 — replaces utils/nn_distance.py:32-59 (and huber_loss :13-30) of the
// reference, which materialises two (B,N,M,3) tensors with .repeat().  Here every query point
// (b,i) is one thread that scans the M points of the other cloud and keeps (min, first argmin);
// nothing larger than the outputs is written.  Call shapes on the hot path are tiny
// ((8192,1,3)x(8192,3,3) L1 and (8,256,3)x(8,256,3) L2^2: lib/loss_helper/loss_detection.py:66,92),
// so the kernel is launch-latency bound.
#include "common.h"

namespace {

template <int MODE>
__device__ __forceinline__ float pair_dist(float ax, float ay, float az, float bx, float by, float bz, float delta) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float e0, e1, e2;
  if (MODE == 0) {  // torch.sum(pc_diff**2, -1)
    e0 = dx * dx; e1 = dy * dy; e2 = dz * dz;
  } else if (MODE == 1) {  // torch.sum(|pc_diff|, -1)
    e0 = fabsf(dx); e1 = fabsf(dy); e2 = fabsf(dz);
  } else {  // huber: 0.5*min(|x|,d)^2 + d*(|x|-min(|x|,d))
    const float a0 = fabsf(dx), a1 = fabsf(dy), a2 = fabsf(dz);
    const float q0 = fminf(a0, delta), q1 = fminf(a1, delta), q2 = fminf(a2, delta);
    const float h0 = 0.5f * (q0 * q0), h1 = 0.5f * (q1 * q1), h2 = 0.5f * (q2 * q2);
    const float l0 = delta * (a0 - q0), l1 = delta * (a1 - q1), l2 = delta * (a2 - q2);
    e0 = h0 + l0; e1 = h1 + l1; e2 = h2 + l2;
  }
  const float s = e0 + e1;
  return s + e2;
}

// For every point i of `a` (B,Na,3): min over the Nb points of `b` of dist(a_i, b_j); first minimum wins.
// SWAP only fixes the subtraction order pc1 - pc2 (irrelevant for the value, kept for bit-fidelity).
template <int MODE, bool SWAP>
__global__ __launch_bounds__(256) void nn_min_kernel(const float *__restrict__ a, const float *__restrict__ bpts,
                                                     long long total, int Na, int Nb, float delta,
                                                     float *__restrict__ dist, long long *__restrict__ idx) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long bi = t / Na;
  const float ax = a[t * 3 + 0], ay = a[t * 3 + 1], az = a[t * 3 + 2];
  const float *__restrict__ q = bpts + bi * Nb * 3;
  float best = __builtin_inff();
  int besti = 0;
  for (int j = 0; j < Nb; ++j) {
    const float d = SWAP ? pair_dist<MODE>(q[j * 3], q[j * 3 + 1], q[j * 3 + 2], ax, ay, az, delta)
                         : pair_dist<MODE>(ax, ay, az, q[j * 3], q[j * 3 + 1], q[j * 3 + 2], delta);
    if (d < best || j == 0) {
      best = d;
      besti = j;
    }
  }
  dist[t] = best;
  idx[t] = besti;
}

template <int MODE>
int launch_mode(const float *pc1, const float *pc2, int B, int N, int M, float delta, float *dist1, long long *idx1,
                float *dist2, long long *idx2, hipStream_t s) {
  const long long t1 = (long long)B * N, t2 = (long long)B * M;
  hipLaunchKernelGGL((nn_min_kernel<MODE, false>), dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, pc1, pc2, t1,
                     N, M, delta, dist1, idx1);
  hipLaunchKernelGGL((nn_min_kernel<MODE, true>), dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, pc2, pc1, t2, M,
                     N, delta, dist2, idx2);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

}  // namespace

extern "C" int vlp3d_nn_distance(const float *pc1, const float *pc2, int B, int N, int M, int mode, float delta,
                                 float *dist1, long long *idx1, float *dist2, long long *idx2, void *stream) {
  if (!pc1 || !pc2 || !dist1 || !idx1 || !dist2 || !idx2 || B < 1 || N < 1 || M < 1 || mode < 0 || mode > 2)
    return VLP3D_EINVAL;
  if ((long long)B * N >= (1ll << 38) || (long long)B * M >= (1ll << 38)) return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) return launch_mode<0>(pc1, pc2, B, N, M, delta, dist1, idx1, dist2, idx2, s);
  if (mode == 1) return launch_mode<1>(pc1, pc2, B, N, M, delta, dist1, idx1, dist2, idx2, s);
  return launch_mode<2>(pc1, pc2, B, N, M, delta, dist1, idx1, dist2, idx2, s);
}
