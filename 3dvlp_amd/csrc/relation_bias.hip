// Pairwise-geometry attention bias of the relation module — replaces, per layer, the reference's
//   weights = cat([centre_j - centre_i, ||centre_j - centre_i||])            (B,K,K,4)
//   dist_weights = self_attn_fc[i](weights).permute(0,3,1,2)                 (B,4,K,K)
// (models/proposal_module/relation_module.py:72-92 with self_attn_fc = Linear(4,32) ReLU LayerNorm(32)
// Linear(32,32) ReLU LayerNorm(32) Linear(32,4), :26-37), which materialises two (B,K,K,32) activations per
// layer (+ LayerNorm statistics) and runs six skinny GEMMs with 524 288 rows through the BLAS library.
//
// Here one thread owns one (b,i,j) pair and carries the whole 4->32->32->4 MLP in registers (weights are
// wave-uniform: scalar loads); nothing but the (B,4,K,K) bias is written.  The backward recomputes the
// forward per pair, back-propagates in registers and reduces the parameter gradients per wave:
// rank-64 updates dW += dZ^T A on the matrix cores (fp32 MFMA, operands read back from a padded LDS tile
// where lane = output feature, k = pair) and column sums for biases / LayerNorm affine parameters.
// The geometry input carries no gradient (the reference detaches it, :86-87).  The file is compiled with
// -ffp-contract=off like the rest of the library; the dot products use explicit fused multiply-adds (as a BLAS GEMM
// does) — half the VALU instructions of separate multiplies and adds.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HID = 32;
constexpr int NPARAM = 4 * HID + HID + HID + HID + HID * HID + HID + HID + HID + 4 * HID + 4;  // 1476
// offsets inside one parameter(-gradient) block
constexpr int O_W1 = 0, O_B1 = 128, O_G1 = 160, O_E1 = 192, O_W2 = 224, O_B2 = 1248, O_G2 = 1280, O_E2 = 1312,
              O_W3 = 1344, O_B3 = 1472;

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Keeps the weight reads of one output feature next to their use: without it the scheduler hoists all 1476 LDS
// reads to the top and holds the weights in 450 VGPRs (one wave per SIMD, scratch spills).
#define ROW_FENCE() asm volatile("" ::: "memory")

// LayerNorm(32) of v (biased variance, eps 1e-5): n = (v-mean)*rstd ; returns rstd
__device__ __forceinline__ float layer_norm32(const float (&v)[HID], float (&n)[HID]) {
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < HID; ++i) mean += v[i];
  mean *= (1.f / HID);
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < HID; ++i) {
    const float d = v[i] - mean;
    var += d * d;
  }
  const float rstd = rsqrtf(var * (1.f / HID) + 1e-5f);
#pragma unroll
  for (int i = 0; i < HID; ++i) n[i] = (v[i] - mean) * rstd;
  return rstd;
}

struct Fwd {
  float x[4];
  unsigned m1, m2;  // relu masks of z1, z2
  float n1[HID], n2[HID];  // h = n*gamma + beta is recomputed where needed (64 fewer live registers in backward)
  float rstd1, rstd2;
  float o[4];
};

__device__ __forceinline__ void pair_input(const float *__restrict__ centre, int b, int i, int j, int K, float (&x)[4]) {
  const float *ci = centre + ((long long)b * K + i) * 3;
  const float *cj = centre + ((long long)b * K + j) * 3;
  x[0] = cj[0] - ci[0];
  x[1] = cj[1] - ci[1];
  x[2] = cj[2] - ci[2];
  const float s = x[0] * x[0] + x[1] * x[1];
  x[3] = sqrtf(s + x[2] * x[2]);
}

template <bool FENCE>
__device__ __forceinline__ void mlp_forward(const float *__restrict__ P, Fwd &f) {
  float r[HID];
  f.m1 = 0u;
#pragma unroll
  for (int i = 0; i < HID; ++i) {
    if (FENCE && (i & 7) == 0) ROW_FENCE();
    float z = P[O_B1 + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_fmaf(P[O_W1 + i * 4 + k], f.x[k], z);
    if (z > 0.f) f.m1 |= 1u << i;
    r[i] = fmaxf(z, 0.f);
  }
  f.rstd1 = layer_norm32(r, f.n1);
  float h1[HID], h2[HID];
#pragma unroll
  for (int i = 0; i < HID; ++i) h1[i] = f.n1[i] * P[O_G1 + i] + P[O_E1 + i];
  f.m2 = 0u;
#pragma unroll
  for (int i = 0; i < HID; ++i) {
    float z = P[O_B2 + i];
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      if (FENCE && (k & 7) == 0) ROW_FENCE();  // at most 8 weights (two ds_read_b128) in flight per thread
      z = __builtin_fmaf(P[O_W2 + i * HID + k], h1[k], z);
    }
    if (z > 0.f) f.m2 |= 1u << i;
    r[i] = fmaxf(z, 0.f);
  }
  f.rstd2 = layer_norm32(r, f.n2);
#pragma unroll
  for (int i = 0; i < HID; ++i) h2[i] = f.n2[i] * P[O_G2 + i] + P[O_E2 + i];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float z = P[O_B3 + c];
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      if (FENCE && (k & 7) == 0) ROW_FENCE();
      z = __builtin_fmaf(P[O_W3 + c * HID + k], h2[k], z);
    }
    f.o[c] = z;
  }
}

__global__ __launch_bounds__(256) void relation_bias_fwd_kernel(const float *__restrict__ centre,
                                                                const float *__restrict__ Pg, int B, int K,
                                                                float *__restrict__ out) {
  // The 1476 weights are read through LDS (same address in every lane: broadcast ds_read_b128).  Reading them as
  // wave-uniform scalars made the compiler load all of them up front and spill the SGPRs through VGPR lanes: 3100
  // v_readlane/v_writelane per forward, one per multiply-add.
  __shared__ __attribute__((aligned(16))) float sp[NPARAM];
  for (int i = threadIdx.x; i < NPARAM; i += 256) sp[i] = Pg[i];
  __syncthreads();
  const float *P = sp;
  const long long total = (long long)B * K * K;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int j = (int)(t % K);
    const int i = (int)((t / K) % K);
    const int b = (int)(t / ((long long)K * K));
    Fwd f;
    pair_input(centre, b, i, j, K, f.x);
    mlp_forward<true>(P, f);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[(((long long)b * 4 + c) * K + i) * K + j] = f.o[c];
  }
}

// LayerNorm backward on 32 features: dn -> d(input of LN) ; n = normalised value
__device__ __forceinline__ void layer_norm32_bwd(const float (&dn)[HID], const float (&n)[HID], float rstd,
                                                 float (&dr)[HID]) {
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int i = 0; i < HID; ++i) {
    m1 += dn[i];
    m2 += dn[i] * n[i];
  }
  m1 *= (1.f / HID);
  m2 *= (1.f / HID);
#pragma unroll
  for (int i = 0; i < HID; ++i) dr[i] = rstd * (dn[i] - m1 - n[i] * m2);
}

constexpr int LDT = HID + 1;  // padded LDS row: lane = pair writes its 32 features without bank conflicts

// acc (32x32) += A^T B over the 64 pairs of a wave tile: A, B are [64][LDT] LDS tiles (row = pair).
__device__ __forceinline__ f32x16 rank64_update(const float *__restrict__ A, const float *__restrict__ Bm, int r,
                                                int half, f32x16 acc) {
#pragma unroll 8
  for (int kk = 0; kk < 32; ++kk) {
    const int row = 2 * kk + half;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[row * LDT + r], Bm[row * LDT + r], acc, 0, 0, 0);
  }
  return acc;
}

// sum over the 64 rows of column (lane & 31); both halves of the wave end up with the full sum
__device__ __forceinline__ float column_sum(const float *__restrict__ A, int r, int half) {
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) s += A[(half * 32 + k) * LDT + r];
  return s + __shfl_xor(s, 32);
}

__global__ __launch_bounds__(256) void relation_bias_bwd_kernel(const float *__restrict__ centre,
                                                                const float *__restrict__ Pg,
                                                                const float *__restrict__ dout, int B, int K,
                                                                float *__restrict__ slabs) {
  extern __shared__ float lds[];  // per wave: two [64][LDT] tiles
  // Backward keeps ~250 values live per thread and stays on wave-uniform SCALAR weights: through LDS (even in fenced
  // chunks of 8) the register allocation collapses to 512 VGPRs + 3.7 KB of scratch per thread (4x slower, measured).
  // The scalar form pays ~4000 v_readlane/v_writelane SGPR spills per 64 pairs instead.
  const float *__restrict__ P = Pg;
  const float *__restrict__ sW2 = Pg + O_W2;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *TA = lds + wave * 2 * 64 * LDT;
  float *TB = TA + 64 * LDT;

  f32x16 accW2, accW3, accW1;
#pragma unroll
  for (int e = 0; e < 16; ++e) accW2[e] = accW3[e] = accW1[e] = 0.f;
  float s_b3 = 0.f, s_g2 = 0.f, s_e2 = 0.f, s_b2 = 0.f, s_g1 = 0.f, s_e1 = 0.f, s_b1 = 0.f;

  const long long total = (long long)B * K * K;
  const long long ntiles = (total + 63) / 64;
  const long long gwave = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
  for (long long tile = gwave; tile < ntiles; tile += nwaves) {
    const long long t = tile * 64 + lane;
    const bool ok = t < total;
    const long long tc = ok ? t : total - 1;
    const int j = (int)(tc % K);
    const int i = (int)((tc / K) % K);
    const int b = (int)(tc / ((long long)K * K));
    Fwd f;
    pair_input(centre, b, i, j, K, f.x);
    mlp_forward<false>(P, f);
    float dO[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dO[c] = ok ? dout[(((long long)b * 4 + c) * K + i) * K + j] : 0.f;

    // ---- layer 3: dW3 (4x32, zero-padded to 32x32) += dO^T h2 ; db3 ; dh2 = W3^T dO
    float d[HID], dn[HID], dr[HID];
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      TA[lane * LDT + k] = k < 4 ? dO[k] : 0.f;
      TB[lane * LDT + k] = f.n2[k] * P[O_G2 + k] + P[O_E2 + k];  // h2
      d[k] = __builtin_fmaf(P[O_W3 + 3 * HID + k], dO[3],
                            __builtin_fmaf(P[O_W3 + 2 * HID + k], dO[2],
                                           __builtin_fmaf(P[O_W3 + HID + k], dO[1], P[O_W3 + k] * dO[0])));
    }
    accW3 = rank64_update(TA, TB, r, half, accW3);
    s_b3 += column_sum(TA, r, half);
    // ---- LN2 affine grads (dg2 = sum dh2*n2, dbe2 = sum dh2), then through LN2 and ReLU
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      TA[lane * LDT + k] = d[k] * f.n2[k];
      TB[lane * LDT + k] = d[k];
      dn[k] = d[k] * P[O_G2 + k];
    }
    s_g2 += column_sum(TA, r, half);
    s_e2 += column_sum(TB, r, half);
    layer_norm32_bwd(dn, f.n2, f.rstd2, dr);
    // ---- layer 2: dz2 ; dW2 += dz2^T h1 ; db2 ; dh1 = W2^T dz2
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      dr[k] = ((f.m2 >> k) & 1u) ? dr[k] : 0.f;
      TA[lane * LDT + k] = dr[k];
      TB[lane * LDT + k] = f.n1[k] * P[O_G1 + k] + P[O_E1 + k];  // h1
    }
    accW2 = rank64_update(TA, TB, r, half, accW2);
    s_b2 += column_sum(TA, r, half);
    // dh1[k] = sum_q W2[q][k] dz2[q]: rows of W2 in the outer loop — contiguous scalar loads (s_load_dwordx8) consumed
    // at once.  The column-wise form (k outer) issued 1024 strided one-dword scalar loads whose results the compiler
    // kept in SGPRs and spilled through VGPR lanes (5700 v_readlane/v_writelane, 256 VGPRs, one wave per SIMD).
    // Same summation order per k (q ascending): bit-identical.
#pragma unroll
    for (int k = 0; k < HID; ++k) d[k] = 0.f;
#pragma unroll
    for (int q = 0; q < HID; ++q) {
#pragma unroll
      for (int k = 0; k < HID; ++k) d[k] = __builtin_fmaf(sW2[q * HID + k], dr[q], d[k]);
    }
    // ---- LN1 affine grads, through LN1 and ReLU
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      TA[lane * LDT + k] = d[k] * f.n1[k];
      TB[lane * LDT + k] = d[k];
      dn[k] = d[k] * P[O_G1 + k];
    }
    s_g1 += column_sum(TA, r, half);
    s_e1 += column_sum(TB, r, half);
    layer_norm32_bwd(dn, f.n1, f.rstd1, dr);
    // ---- layer 1: dz1 ; dW1 (32x4, zero-padded) += dz1^T x ; db1
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      TA[lane * LDT + k] = ((f.m1 >> k) & 1u) ? dr[k] : 0.f;
      TB[lane * LDT + k] = k < 4 ? f.x[k] : 0.f;
    }
    accW1 = rank64_update(TA, TB, r, half, accW1);
    s_b1 += column_sum(TA, r, half);
  }

  // ---- this wave's partial parameter gradients -> its slab (summed on the host side of the ABI call)
  float *S = slabs + gwave * NPARAM;
  // accumulators: element (row = acc_row(e,half) = A-feature, col = r = B-feature)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int a = acc_row(e, half);
    S[O_W2 + a * HID + r] = accW2[e];                 // dW2[i=a][k=r]
    if (a < 4) S[O_W3 + a * HID + r] = accW3[e];      // dW3[c=a][k=r]
    if (r < 4) S[O_W1 + a * 4 + r] = accW1[e];        // dW1[i=a][k=r]
  }
  if (half == 0) {
    if (r < 4) S[O_B3 + r] = s_b3;
    S[O_G2 + r] = s_g2; S[O_E2 + r] = s_e2; S[O_B2 + r] = s_b2;
    S[O_G1 + r] = s_g1; S[O_E1 + r] = s_e1; S[O_B1 + r] = s_b1;
  }
}

// out[i] = sum_k slabs[k][i]: 256 threads = 4 slab-groups x 64 consecutive elements
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ slabs, int nslab, int n,
                                                       float *__restrict__ out) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  float s = 0.f;
  if (i < n)
    for (int k = grp; k < nslab; k += 4) s += slabs[(long long)k * n + i];
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i < n) out[i] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

}  // namespace

extern "C" int vlp3d_relation_bias_nparam(void) { return NPARAM; }

// params: one block of NPARAM floats = [W1(32x4) b1 g1 be1 W2(32x32) b2 g2 be2 W3(4x32) b3]
// centre (B,K,3) -> out (B,4,K,K)
extern "C" int vlp3d_relation_bias_fwd(const float *centre, const float *params, int B, int K, float *out,
                                       void *stream) {
  if (!centre || !params || !out || B < 1 || K < 1) return VLP3D_EINVAL;
  const long long total = (long long)B * K * K;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(relation_bias_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, centre, params,
                     B, K, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// dout (B,4,K,K) -> dparams (NPARAM); slabs: scratch of nblocks*4*NPARAM floats (nblocks >= 1).
extern "C" int vlp3d_relation_bias_bwd(const float *centre, const float *params, const float *dout, int B, int K,
                                       float *dparams, float *slabs, int nblocks, void *stream) {
  if (!centre || !params || !dout || !dparams || !slabs || B < 1 || K < 1 || nblocks < 1) return VLP3D_EINVAL;
  const long long ntiles = ((long long)B * K * K + 63) / 64;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > nblocks) blocks = nblocks;
  const size_t lds = (size_t)4 * 2 * 64 * LDT * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(relation_bias_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, s, centre, params, dout, B, K,
                     slabs);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((NPARAM + 63) / 64), dim3(256), 0, s, slabs, (int)blocks * 4, NPARAM,
                     dparams);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
