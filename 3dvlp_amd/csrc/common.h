// Shared device helpers for the gfx950 kernels of libvlp3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlp3d.h"

#ifndef VLP3D_CONTRACT
#define VLP3D_CONTRACT 1
#endif

#define VLP3D_WAVE 64

// (a*a)+(b*b)+(c*c) in the evaluation order of the reference build (vlp3d.h: vlp3d_fp_contract).
// The file is compiled with -ffp-contract=off; every fused multiply-add is explicit.
__device__ __forceinline__ float vlp3d_sumsq3(float a, float b, float c) {
#if VLP3D_CONTRACT == 1
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#elif VLP3D_CONTRACT == 2
  return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#else
  float t = a * a;
  float u = b * b;
  float s = t + u;
  float v = c * c;
  return s + v;
#endif
}

// p1*w1 + p2*w2 + p3*w3 in the same evaluation order (interpolate_gpu.cu:103-104).
__device__ __forceinline__ float vlp3d_blend3(float p1, float w1, float p2, float w2, float p3, float w3) {
#if VLP3D_CONTRACT == 1
  return __builtin_fmaf(p3, w3, __builtin_fmaf(p1, w1, p2 * w2));
#elif VLP3D_CONTRACT == 2
  return __builtin_fmaf(p3, w3, __builtin_fmaf(p2, w2, p1 * w1));
#else
  float a = p1 * w1;
  float b = p2 * w2;
  float s = a + b;
  float c = p3 * w3;
  return s + c;
#endif
}

// sampling_gpu.cu:105-106: `mag <= 1e-3` compares the float against a DOUBLE literal.
__device__ __forceinline__ bool vlp3d_fps_skipped(float x, float y, float z) {
  return (double)vlp3d_sumsq3(x, y, z) <= 1e-3;
}

__host__ __device__ __forceinline__ int vlp3d_cdiv(int a, int b) { return (a + b - 1) / b; }

// Counter-based dropout masks (csrc/add_norm.hip, csrc/caption.hip): a hash of (seed word, call id, element index),
// recomputed bit for bit in backward, never stored.
static __device__ __forceinline__ unsigned pcg_hash(unsigned x) {
  x = x * 747796405u + 2891336453u;
  const unsigned w = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
  return (w >> 22u) ^ w;
}

// keep-probability test of element `e` of call `call_id` under `seed`; 24 random bits against the threshold
static __device__ __forceinline__ bool keep_element(unsigned seed_mix, unsigned e, unsigned thresh24) {
  return (pcg_hash(e ^ seed_mix) >> 8) >= thresh24;
}

static __device__ __forceinline__ unsigned seed_mix_of(const unsigned long long *__restrict__ seed, int call_id) {
  const unsigned long long s = *seed;
  return pcg_hash((unsigned)s ^ pcg_hash((unsigned)(s >> 32) + 0x9E3779B9u * (unsigned)(call_id + 1)));
}

// Opt a kernel in to more than 64 KB of dynamic LDS — once per DEVICE (the attribute is per device; a process-wide flag left a
// second GPU without it: ADVICE r3).  `done` is one word per call site: bit d = device d has the attribute.
#include <atomic>
static inline int vlp3d_opt_in_lds(const void *fn, int bytes, std::atomic<unsigned long long> &done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VLP3D_EINVAL;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return VLP3D_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return VLP3D_OK;
}

#define VLP3D_LAUNCH_CHECK()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

// Clear n 32-bit words from inside the stream's kernel order.  Used instead of hipMemsetAsync everywhere in this library:
// captured into a HIP graph (ROCm 7.2), memset nodes were observed NOT to be ordered against the neighbouring kernel
// nodes / the previous replay — an accumulate-into-zeroed-buffer kernel saw the old contents of the block on the second
// replay (tools/memset_graph_probe.py; a ball-query counter array left uncleared that way ended in an out-of-bounds write).
static __global__ __launch_bounds__(256) void vlp3d_zero_words_kernel(uint32_t *__restrict__ p, size_t n) {
  const size_t head = ((16 - ((uintptr_t)p & 15)) & 15) >> 2;  // words in front of the first 16-byte boundary
  const size_t h = head < n ? head : n;
  const size_t vec = (n - h) >> 2;
  uint4 *v = reinterpret_cast<uint4 *>(p + h);
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (size_t i = tid; i < vec; i += stride) v[i] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < h) p[tid] = 0u;
  const size_t tail0 = h + (vec << 2);
  if (tid < n - tail0) p[tail0 + tid] = 0u;
}

static inline hipError_t vlp3d_zero_words(void *p, size_t n_words, hipStream_t s) {
  if (n_words == 0) return hipSuccess;
  size_t blocks = (n_words / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(vlp3d_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t *)p, n_words);
  return hipGetLastError();
}
