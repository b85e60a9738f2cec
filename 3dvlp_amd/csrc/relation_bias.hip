// Pairwise-geometry attention bias of the relation module — replaces, per layer, the reference's
//   weights = cat([centre_j - centre_i, ||centre_j - centre_i||])            (B,K,K,4)
//   dist_weights = self_attn_fc[i](weights).permute(0,3,1,2)                 (B,4,K,K)
// (models/proposal_module/relation_module.py:72-92 with self_attn_fc = Linear(4,32) ReLU LayerNorm(32)
// Linear(32,32) ReLU LayerNorm(32) Linear(32,4), :26-37), which materialises two (B,K,K,32) activations per
// layer (+ LayerNorm statistics) and runs six skinny GEMMs with 524 288 rows through the BLAS library.
//
// Here one thread owns one (b,i,j) pair and carries the whole 4->32->32->4 MLP in registers (weights are
// read through LDS); nothing but the (B,4,K,K) bias is written.  The backward recomputes the forward per pair with
// two lanes per pair (16 features each), runs the two 32x32 layer products on the matrix cores, and reduces the
// parameter gradients per wave: rank-32 updates dW += dZ^T A (fp32 MFMA, operands read back from a padded LDS tile
// where lane = output feature, k = pair) and column sums for biases / LayerNorm affine parameters.
// The geometry input carries no gradient (the reference detaches it, :86-87).  The file is compiled with
// -ffp-contract=off like the rest of the library; the dot products use explicit fused multiply-adds (as a BLAS GEMM
// does) — half the VALU instructions of separate multiplies and adds.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HID = 32;
constexpr int NPARAM = 4 * HID + HID + HID + HID + HID * HID + HID + HID + HID + 4 * HID + 4;  // 1476
static_assert(NPARAM % 4 == 0, "slab_sum_kernel reads the slabs as float4");
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
    const unsigned tu = (unsigned)t;  // B*K*K < 2^32 (host-checked): 32-bit divisions
    const int j = (int)(tu % (unsigned)K);
    const int i = (int)((tu / (unsigned)K) % (unsigned)K);
    const int b = (int)(tu / ((unsigned)K * (unsigned)K));
    Fwd f;
    pair_input(centre, b, i, j, K, f.x);
    mlp_forward<true>(P, f);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[(((long long)b * 4 + c) * K + i) * K + j] = f.o[c];
  }
}

constexpr int LDT = HID + 1;  // padded LDS row: lane = pair writes its 32 features without bank conflicts

// ------------------------------------------------------------------------------------------------------------------
// Backward: TWO lanes per pair.  Lane (j = lane & 31, h = lane >> 5) of a wave holds, for pair j of the
// current 32-pair tile, the 16 features F(e,h) = acc_row(e,h) — exactly the rows an MFMA accumulator lane holds.
// That halves the per-thread state (a one-lane-per-pair backward needs ~250 live values and spilled its scalar
// weights through VGPR lanes: 278 us per layer against 202 us here) and puts the two 32x32 layer products on the matrix cores in exact fp32:
//   Z2^T (feature x pair) = W2 * H1^T        A operand = W2 columns permuted to F(kk,h), B operand = h1[kk] (registers)
//   dH1^T                 = W2^T * dZ2^T     A operand = W2 rows    permuted to F(kk,h), B operand = dz2[kk]
// the output lands in the same split layout (accumulator-as-next-operand, as in sdpa.hip) — no transposes.  Per-pair
// LayerNorm statistics need one exchange with lane ^ 32.  Parameter gradients as before: rank-32 MFMA updates over
// the pairs of a tile (operands through a padded LDS tile) + column sums, per-wave slabs.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 rank32_update(const float *__restrict__ A, const float *__restrict__ Bm, int r, int half,
                                                f32x16 acc) {
#pragma unroll 8
  for (int kk = 0; kk < 16; ++kk) {
    const int row = 2 * kk + half;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[row * LDT + r], Bm[row * LDT + r], acc, 0, 0, 0);
  }
  return acc;
}
typedef short bf16x8_rb __attribute__((ext_vector_type(8)));
__device__ __forceinline__ short rb_bf16(float v) {
  __hip_bfloat16 h = __float2bfloat16(v);
  return *reinterpret_cast<short *>(&h);
}
// bf16 form of rank32_update (timing configuration): the same contraction over the tile's 32 pairs as two 32x32x16 steps;
// lane (r, half) supplies pairs 16 s + 8 half + t, t = 0..7, of feature column r (operands rounded to bf16, fp32 accumulation)
__device__ __forceinline__ f32x16 rank32_update_bf16(const float *__restrict__ A, const float *__restrict__ Bm, int r, int half,
                                                     f32x16 acc) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    bf16x8_rb a, b;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = 16 * s2 + 8 * half + t;
      a[t] = rb_bf16(A[row * LDT + r]);
      b[t] = rb_bf16(Bm[row * LDT + r]);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ float column_sum32(const float *__restrict__ A, int r, int half) {
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < 16; ++k) s += A[(half * 16 + k) * LDT + r];
  return s + __shfl_xor(s, 32);
}

// BF (the step's bf16 timing configuration): the input-gradient product dH1 = W2^T dZ2 and the three rank-32 parameter-gradient
// updates run as v_mfma_f32_32x32x16_bf16 with operands rounded to bf16 in registers (16 exact + 8 bf16 matrix instructions per
// 32-pair tile instead of 80 exact ones: the fp32 MFMAs were a third of the kernel); the recomputed forward (layer 1, the
// layer-2 product, both LayerNorms) and all sums stay fp32.
template <bool BF, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void relation_bias_bwd_kernel(const float *__restrict__ centre,
                                                                 const float *__restrict__ Pg,
                                                                 const float *__restrict__ dout, int B, int K,
                                                                 float *__restrict__ slabs) {
  extern __shared__ float lds[];
  float *sp = lds;                   // [NPARAM] plain parameters
  float *A2 = sp + NPARAM + 4;       // [(h*16+kk)*32 + i] = W2[i][F(kk,h)]
  float *A2T = A2 + HID * HID;       // [(h*16+kk)*32 + k] = W2[F(kk,h)][k]
  float *tiles = A2T + HID * HID;    // per wave: two [32][LDT] tiles
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *TA = tiles + wave * 2 * 32 * LDT;
  float *TB = TA + 32 * LDT;
  for (int i = threadIdx.x; i < NPARAM; i += 256) sp[i] = Pg[i];
  for (int i = threadIdx.x; i < HID * HID; i += 256) {
    const int hh = i >> 9, kk = (i >> 5) & 15, c = i & 31;
    const int f = acc_row(kk, hh);
    A2[i] = Pg[O_W2 + c * HID + f];
    A2T[i] = Pg[O_W2 + f * HID + c];
  }
  // bf16 image for the 32x32x16 input-gradient product: [(s2 * 2 + h) * 32 + i][8]: element t = W2[F(8 s2 + t, h)][i] — one
  // 16-byte LDS read per operand
  short *A2Tb = reinterpret_cast<short *>(tiles + 4 * 2 * 32 * LDT);
  if (BF) {
    for (int i = threadIdx.x; i < HID * HID; i += 256) {
      const int t = i & 7, row = (i >> 3) & 31, sh = i >> 8;  // sh = s2 * 2 + h
      A2Tb[i] = rb_bf16(Pg[O_W2 + acc_row(8 * (sh >> 1) + t, sh & 1) * HID + row]);
    }
  }
  __syncthreads();
  const int fo = 4 * half;  // F(e,h) = (e & 3) + 8 * (e >> 2) + 4h: compile-time part + fo
#define FEAT(e) (((e) & 3) + 8 * ((e) >> 2))

  f32x16 accW2, accW3, accW1;
#pragma unroll
  for (int e = 0; e < 16; ++e) accW2[e] = accW3[e] = accW1[e] = 0.f;
  float s_b3 = 0.f, s_g2 = 0.f, s_e2 = 0.f, s_b2 = 0.f, s_g1 = 0.f, s_e1 = 0.f, s_b1 = 0.f;

  const long long total = (long long)B * K * K;
  const long long ntiles = (total + 31) / 32;
  const long long gwave = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
  for (long long tile = gwave; tile < ntiles; tile += nwaves) {
    const long long t = tile * 32 + r;
    const bool ok = t < total;
    const long long tc = ok ? t : total - 1;
    const unsigned tu = (unsigned)tc;  // B*K*K < 2^32 (host-checked): 32-bit divisions
    const int j = (int)(tu % (unsigned)K);
    const int i = (int)((tu / (unsigned)K) % (unsigned)K);
    const int b = (int)(tu / ((unsigned)K * (unsigned)K));
    float x[4];
    pair_input(centre, b, i, j, K, x);
    // ---- forward, layer 1 (K = 4: vector unit) + LN1
    float n1[16], n2[16], v[16];
    unsigned m1 = 0u, m2 = 0u;
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = FEAT(e) + fo;
      const float4 w = *reinterpret_cast<const float4 *>(sp + O_W1 + 4 * f);
      float z = sp[O_B1 + f];
      z = __builtin_fmaf(w.x, x[0], z); z = __builtin_fmaf(w.y, x[1], z);
      z = __builtin_fmaf(w.z, x[2], z); z = __builtin_fmaf(w.w, x[3], z);
      if (z > 0.f) m1 |= 1u << e;
      v[e] = fmaxf(z, 0.f);
      sum += v[e];
    }
    sum += __shfl_xor(sum, 32);
    float mean = sum * (1.f / HID), var = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float d0 = v[e] - mean;
      var += d0 * d0;
    }
    var += __shfl_xor(var, 32);
    const float rstd1 = rsqrtf(var * (1.f / HID) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 16; ++e) n1[e] = (v[e] - mean) * rstd1;
    // ---- layer 2 on the matrix cores: Z2^T = W2 H1^T
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // (also in the bf16 configuration this product stays exact: z2 decides the ReLU mask and feeds LayerNorm 2's statistics,
    // and a bf16 z2 put a coherent 3 % error on every gradient upstream of it — tools/relbias_bf16_err.py)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int f = FEAT(kk) + fo;
      const float h1 = __builtin_fmaf(n1[kk], sp[O_G1 + f], sp[O_E1 + f]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[(half * 16 + kk) * 32 + r], h1, acc, 0, 0, 0);
    }
    sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float z = acc[e] + sp[O_B2 + FEAT(e) + fo];
      if (z > 0.f) m2 |= 1u << e;
      v[e] = fmaxf(z, 0.f);
      sum += v[e];
    }
    sum += __shfl_xor(sum, 32);
    mean = sum * (1.f / HID);
    var = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float d0 = v[e] - mean;
      var += d0 * d0;
    }
    var += __shfl_xor(var, 32);
    const float rstd2 = rsqrtf(var * (1.f / HID) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 16; ++e) n2[e] = (v[e] - mean) * rstd2;

    // ---- backward.  dO of this lane's pair
    float dO[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dO[c] = ok ? dout[(((long long)b * 4 + c) * K + i) * K + j] : 0.f;
    // layer 3: dW3 (4x32, zero-padded) += dO^T h2 ; db3 ; dh2 = W3^T dO
    float d[16], dn[16];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = FEAT(e) + fo;
      TA[r * LDT + f] = (half == 0 && e < 4) ? dO[e & 3] : 0.f;  // feature f < 4 only for (h = 0, e < 4), where f = e
      TB[r * LDT + f] = __builtin_fmaf(n2[e], sp[O_G2 + f], sp[O_E2 + f]);  // h2
      d[e] = __builtin_fmaf(sp[O_W3 + 3 * HID + f], dO[3],
                            __builtin_fmaf(sp[O_W3 + 2 * HID + f], dO[2],
                                           __builtin_fmaf(sp[O_W3 + HID + f], dO[1], sp[O_W3 + f] * dO[0])));
    }
    accW3 = BF ? rank32_update_bf16(TA, TB, r, half, accW3) : rank32_update(TA, TB, r, half, accW3);
    s_b3 += column_sum32(TA, r, half);
    // LN2 affine grads, then through LN2 and ReLU
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = FEAT(e) + fo;
      TA[r * LDT + f] = d[e] * n2[e];
      TB[r * LDT + f] = d[e];
      dn[e] = d[e] * sp[O_G2 + f];
      a1 += dn[e];
      a2 += dn[e] * n2[e];
    }
    s_g2 += column_sum32(TA, r, half);
    s_e2 += column_sum32(TB, r, half);
    a1 += __shfl_xor(a1, 32);
    a2 += __shfl_xor(a2, 32);
    a1 *= (1.f / HID);
    a2 *= (1.f / HID);
    // layer 2: dz2 ; dW2 += dz2^T h1 ; db2
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = FEAT(e) + fo;
      const float g = rstd2 * (dn[e] - a1 - n2[e] * a2);
      v[e] = ((m2 >> e) & 1u) ? g : 0.f;  // dz2
      TA[r * LDT + f] = v[e];
      TB[r * LDT + f] = __builtin_fmaf(n1[e], sp[O_G1 + f], sp[O_E1 + f]);  // h1
    }
    accW2 = BF ? rank32_update_bf16(TA, TB, r, half, accW2) : rank32_update(TA, TB, r, half, accW2);
    s_b2 += column_sum32(TA, r, half);
    // dH1^T = W2^T dZ2^T on the matrix cores
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (BF) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8_rb vb;
#pragma unroll
        for (int t = 0; t < 8; ++t) vb[t] = rb_bf16(v[8 * s2 + t]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_rb *>(A2Tb + ((s2 * 2 + half) * 32 + r) * 8), vb,
                                                      acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A2T[(half * 16 + kk) * 32 + r], v[kk], acc, 0, 0, 0);
    }
    // LN1 affine grads, through LN1 and ReLU
    a1 = a2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = FEAT(e) + fo;
      d[e] = acc[e];
      TA[r * LDT + f] = d[e] * n1[e];
      TB[r * LDT + f] = d[e];
      dn[e] = d[e] * sp[O_G1 + f];
      a1 += dn[e];
      a2 += dn[e] * n1[e];
    }
    s_g1 += column_sum32(TA, r, half);
    s_e1 += column_sum32(TB, r, half);
    a1 += __shfl_xor(a1, 32);
    a2 += __shfl_xor(a2, 32);
    a1 *= (1.f / HID);
    a2 *= (1.f / HID);
    // layer 1: dz1 ; dW1 (32x4, zero-padded) += dz1^T x ; db1
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = FEAT(e) + fo;
      const float g = rstd1 * (dn[e] - a1 - n1[e] * a2);
      TA[r * LDT + f] = ((m1 >> e) & 1u) ? g : 0.f;
      TB[r * LDT + f] = (half == 0 && e < 4) ? x[e & 3] : 0.f;
    }
    accW1 = BF ? rank32_update_bf16(TA, TB, r, half, accW1) : rank32_update(TA, TB, r, half, accW1);
    s_b1 += column_sum32(TA, r, half);
  }
#undef FEAT

  // ---- the four waves' partial parameter gradients are combined through LDS: ONE slab per workgroup
  __syncthreads();  // every wave is done with the staged weights and its tiles
  float *S = lds + wave * NPARAM;
  for (int i = lane; i < NPARAM; i += 64) S[i] = 0.f;  // entries the scatter below does not touch (none today)
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
  __syncthreads();
  float *G = slabs + (long long)blockIdx.x * NPARAM;
  for (int i = threadIdx.x; i < NPARAM; i += 256)
    G[i] = (lds[i] + lds[NPARAM + i]) + (lds[2 * NPARAM + i] + lds[3 * NPARAM + i]);
}

// out[i] = sum_k slabs[k][i]: a block sums 64 consecutive elements — 16 threads x float4 — in 16 slab-groups, then folds
// the groups through LDS (n % 4 == 0).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ slabs, int nslab, int n,
                                                       float *__restrict__ out) {
  __shared__ float4 red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 64 + 4 * q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
#pragma unroll 4
    for (int k = grp; k < nslab; k += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(slabs + (long long)k * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[grp][q] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float4 t = red[0][q];
#pragma unroll
    for (int g = 1; g < 16; ++g) {
      const float4 v = red[g][q];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + i) = t;
  }
}

}  // namespace

extern "C" int vlp3d_relation_bias_nparam(void) { return NPARAM; }

// params: one block of NPARAM floats = [W1(32x4) b1 g1 be1 W2(32x32) b2 g2 be2 W3(4x32) b3]
// centre (B,K,3) -> out (B,4,K,K)
extern "C" int vlp3d_relation_bias_fwd(const float *centre, const float *params, int B, int K, float *out,
                                       void *stream) {
  if (!centre || !params || !out || B < 1 || K < 1 || (long long)B * K * K >= (1ll << 32)) return VLP3D_EINVAL;
  const long long total = (long long)B * K * K;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(relation_bias_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, centre, params,
                     B, K, out);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}

// dout (B,4,K,K) -> dparams (NPARAM); slabs: scratch of at least nblocks*NPARAM floats (nblocks >= 1).
extern "C" int vlp3d_relation_bias_bwd(const float *centre, const float *params, const float *dout, int B, int K,
                                       float *dparams, float *slabs, int nblocks, int bf16_mma, void *stream) {
  if (!centre || !params || !dout || !dparams || !slabs || B < 1 || K < 1 || nblocks < 1 ||
      (long long)B * K * K >= (1ll << 32))
    return VLP3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long long ntiles = ((long long)B * K * K + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > nblocks) blocks = nblocks;
  const size_t lds = (size_t)(NPARAM + 4 + 2 * HID * HID + 4 * 2 * 32 * LDT) * sizeof(float) + HID * HID * sizeof(short);
  static const int minb = getenv("VLP3D_RELBIAS_MINB") ? atoi(getenv("VLP3D_RELBIAS_MINB")) : 2;
  if (bf16_mma && minb == 1)
    hipLaunchKernelGGL((relation_bias_bwd_kernel<true, 1>), dim3((unsigned)blocks), dim3(256), lds, s, centre, params, dout, B, K,
                       slabs);
  else if (bf16_mma)
    hipLaunchKernelGGL(relation_bias_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, centre, params, dout, B, K,
                       slabs);
  else
    hipLaunchKernelGGL(relation_bias_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, centre, params, dout, B, K,
                       slabs);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((NPARAM + 63) / 64), dim3(256), 0, s, slabs, (int)blocks, NPARAM, dparams);
  VLP3D_LAUNCH_CHECK();
  return VLP3D_OK;
}
