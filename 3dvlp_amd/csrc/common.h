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

#define VLP3D_LAUNCH_CHECK()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)
