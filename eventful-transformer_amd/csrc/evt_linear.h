// evt_linear.h -- types and helpers of the gated-linear kernels (evt_linear.hip).
#pragma once
#include "evt_common.h"

struct LinArgs {
  const float* A; int64_t lda; const int32_t* a_idx; int a_rows;
  const float* W; const uint16_t* Wsplit; const float* bias;
  float* out; int64_t ldo; const int32_t* o_idx; int o_rows;
  const int32_t* count; float* p_upd;
  int B, kcap, K, Nout, act;
  float* ws; int64_t ws_bytes;
  // evt_linear_pipe.hip only (set by evt_gated_mlp for its hidden scratch): A / out hold hl32 lines instead of fp32
  int a_planes, out_planes;
  // evt_linear_pipe.hip only: A is ONE bf16 plane (row pitch lda elements) of exactly bf16-representable values (ABI 4: a_bf16)
  int a_bf16;
};

// evt_linear_pipe.hip: persistent 256-row tiles, software-pipelined k-tiles, for launches that fill the chip.  evt_big_choice: the
// tile configuration the launch would get (0 = none: run the 128x128 kernel); evt_launch_split_big launches it (false = not taken).
int evt_big_choice(const LinArgs& a);
bool evt_launch_split_big(const LinArgs& a, hipStream_t s);

// evt_linear_small.hip: latency-oriented kernel for small gated row counts (one video stream).  Returns the K split it
// launched with (0 = not taken; > 1 = partial planes in the workspace, the caller runs the finish pass).
int evt_launch_split_small(const LinArgs& a, hipStream_t s);
// true: evt_launch_split_small would take this launch (shape-only)
bool evt_small_accepts(const LinArgs& a);

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// GELU with the exact-erf definition (nn.GELU(), blocks.py:114): 0.5 x (1 + erf(x / sqrt 2)).  erf by the rational
// approximation of Abramowitz & Stegun 7.1.28, erf z = 1 - (1 + a1 z + ... + a6 z^6)^-16 for z >= 0 (|error| < 3e-7 in exact
// arithmetic, < 2e-6 as evaluated in fp32, so GELU(x) is off by at most ~1e-6 |x| -- below the 1e-5 of the split-precision
// product that feeds it): 6 FMAs, 4 squarings, one reciprocal, no branches, about a third of the instructions of the device
// library's erff.  The epilogue of the 256-row kernel is not hidden behind other workgroups' MFMAs: with erff the MLP-1
// launch spent a quarter of its time there (474 -> 454 us at B = 256 with this form).  -DEVT_EXACT_ERF builds the erff form.
// Used by the split-precision kernels only; the fp32-MFMA kernel calls gelu_erf_exact below.
__device__ __forceinline__ float gelu_erf(float x) {
#ifndef EVT_EXACT_ERF
  const float z = fabsf(x) * 0.70710678118654752440f;
  float p = 0.0000430638f;
  p = fmaf(p, z, 0.0002765672f);
  p = fmaf(p, z, 0.0001520143f);
  p = fmaf(p, z, 0.0092705272f);
  p = fmaf(p, z, 0.0422820123f);
  p = fmaf(p, z, 0.0705230784f);
  p = fmaf(p, z, 1.0f);
  p *= p; p *= p; p *= p; p *= p;
  const float e = 1.0f - __builtin_amdgcn_rcpf(p);          // erf(|x| / sqrt 2); p >= 1
  return 0.5f * x * (1.0f + copysignf(e, x));
#else
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#endif
}
// Two values at once on the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): 21 VALU instructions per pair instead
// of ~18 per value, the same IEEE operations in the same order (bit-identical to gelu_erf).  For epilogues that are VALU-bound with no
// MFMA beside them (evt_linear_pipe.hip: the GELU of MLP-1 was 14 % of that launch).
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
#ifndef EVT_EXACT_ERF
  const f32x2_t z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  f32x2_t p = {0.0000430638f, 0.0000430638f};
  p = __builtin_elementwise_fma(p, z, (f32x2_t){0.0002765672f, 0.0002765672f});
  p = __builtin_elementwise_fma(p, z, (f32x2_t){0.0001520143f, 0.0001520143f});
  p = __builtin_elementwise_fma(p, z, (f32x2_t){0.0092705272f, 0.0092705272f});
  p = __builtin_elementwise_fma(p, z, (f32x2_t){0.0422820123f, 0.0422820123f});
  p = __builtin_elementwise_fma(p, z, (f32x2_t){0.0705230784f, 0.0705230784f});
  p = __builtin_elementwise_fma(p, z, (f32x2_t){1.0f, 1.0f});
  p *= p; p *= p; p *= p; p *= p;
  f32x2_t e = {1.0f - __builtin_amdgcn_rcpf(p.x), 1.0f - __builtin_amdgcn_rcpf(p.y)};
  e = __builtin_elementwise_copysign(e, x);
  return (x * 0.5f) * (e + 1.0f);
#else
  return (f32x2_t){gelu_erf(x.x), gelu_erf(x.y)};
#endif
}
// The exact-fp32 arithmetic mode (EVT_GEMM=f32: gated_linear_kernel, v_mfma_f32_32x32x2_f32) keeps the reference's GELU, erff.
__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------------------------
// Split weights, "hl32" layout (evt_split_weights): every fp32 weight w is written w = hi + lo + O(2^-17 |w|) with
// hi = rne_bf16(w), lo = rne_bf16(w - hi).  A row of K weights is stored as ceil(K / 32) groups of 128 bytes:
// [32 x bf16 hi | 32 x bf16 lo] of 32 consecutive k (zero-filled past K), so one 32-wide k-tile of a row is ONE
// aligned 128-byte line holding both planes.  Row pitch in bf16 elements:
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int64_t hl32_pitch(int K) { return (int64_t)((K + 31) / 32) * 64; }
// element offset (bf16 units, relative to the row) of the hi value of column k; the lo value sits 32 elements later
__host__ __device__ inline int hl32_hi(int k) { return (k >> 5) * 64 + (k & 31); }

// hi = rne_bf16(x), lo = rne_bf16(x - hi) for 4 values.  Written pairwise so that the fp32 image of hi comes from the
// PACKED conversion by a shift / mask (12 VALU per float4: 4 v_cvt_pk, 2 shifts, 2 ands, 4 subs) instead of hipcc's
// four extra single-element conversions (16).
__device__ __forceinline__ void split4(const float4 v, bf16x4_t* hi, bf16x4_t* lo) {
  union { bf16x2_t b; uint32_t u; } h01, h23, l01, l23;
  h01.b = __builtin_convertvector((f32x2_t){v.x, v.y}, bf16x2_t);
  h23.b = __builtin_convertvector((f32x2_t){v.z, v.w}, bf16x2_t);
  const float r0 = v.x - __uint_as_float(h01.u << 16), r1 = v.y - __uint_as_float(h01.u & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(h23.u << 16), r3 = v.w - __uint_as_float(h23.u & 0xffff0000u);
  l01.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
  l23.b = __builtin_convertvector((f32x2_t){r2, r3}, bf16x2_t);
  union { uint2 u; bf16x4_t b; } H, L;
  H.u = make_uint2(h01.u, h23.u);
  L.u = make_uint2(l01.u, l23.u);
  *hi = H.b;
  *lo = L.b;
}
