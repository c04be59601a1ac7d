// evt_attn_tiles.h -- LDS tile shapes and MFMA sweeps shared by the fused gated-attention kernels
// (evt_attn_fused.hip: N x N score state or N <= 256 in-kernel scores; evt_attn_stream.hip: any N, in-kernel scores).
#pragma once
#include "evt_linear.h"   // split4 (fp32 -> bf16 hi / lo), bf16x8_t

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

constexpr int FR = 32;    // rows per workgroup
constexpr int FKC = 64;   // selected columns per chunk

template <typename T> struct Tile;  // LDS row pitch (elements) and the MFMA sweep over one chunk
template <> struct Tile<bf16_t> {
  static constexpr int PITCH = FKC + 8;
  static __device__ __forceinline__ f32x16 sweep(const bf16_t* a, const bf16_t* b, int lh, f32x16 acc) {
#pragma unroll
    for (int kk = 0; kk < FKC; kk += 16) {
      const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(a + kk + 8 * lh);
      const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(b + kk + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
  // the same over the k range [KB, KB + KLEN) of the chunk only
  template <int KB, int KLEN>
  static __device__ __forceinline__ f32x16 sweep_k(const bf16_t* a, const bf16_t* b, int lh, f32x16 acc) {
#pragma unroll
    for (int kk = KB; kk < KB + KLEN; kk += 16) {
      const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(a + kk + 8 * lh);
      const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(b + kk + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
};
template <> struct Tile<f16_t> {
  static constexpr int PITCH = FKC + 8;
  static __device__ __forceinline__ f32x16 sweep(const f16_t* a, const f16_t* b, int lh, f32x16 acc) {
#pragma unroll
    for (int kk = 0; kk < FKC; kk += 16) {
      const f16x8_t fa = *reinterpret_cast<const f16x8_t*>(a + kk + 8 * lh);
      const f16x8_t fb = *reinterpret_cast<const f16x8_t*>(b + kk + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
  template <int KB, int KLEN>
  static __device__ __forceinline__ f32x16 sweep_k(const f16_t* a, const f16_t* b, int lh, f32x16 acc) {
#pragma unroll
    for (int kk = KB; kk < KB + KLEN; kk += 16) {
      const f16x8_t fa = *reinterpret_cast<const f16x8_t*>(a + kk + 8 * lh);
      const f16x8_t fb = *reinterpret_cast<const f16x8_t*>(b + kk + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
};
template <> struct Tile<float> {
  static constexpr int PITCH = FKC + 4;
  static __device__ __forceinline__ f32x16 sweep(const float* a, const float* b, int lh, f32x16 acc) {
    // permuted k: lane half lh covers k in [64*lh, 64*lh + 64)
#pragma unroll
    for (int q = 0; q < FKC / 2; q += 4) {
      const float4 fa = *reinterpret_cast<const float4*>(a + lh * (FKC / 2) + q);
      const float4 fb = *reinterpret_cast<const float4*>(b + lh * (FKC / 2) + q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
    }
    return acc;
  }
  template <int KB, int KLEN>   // permuted k: lane half lh covers [KB + lh * KLEN / 2, + KLEN / 2)
  static __device__ __forceinline__ f32x16 sweep_k(const float* a, const float* b, int lh, f32x16 acc) {
#pragma unroll
    for (int q = 0; q < KLEN / 2; q += 4) {
      const float4 fa = *reinterpret_cast<const float4*>(a + KB + lh * (KLEN / 2) + q);
      const float4 fb = *reinterpret_cast<const float4*>(b + KB + lh * (KLEN / 2) + q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
    }
    return acc;
  }
};

// exp via v_exp_f32 (2^x): |rel err| ~ 1e-7 * (1 + |x|), far inside the 1e-3 activation tolerance and
// applied to BOTH the normaliser and the gathered numerators, so rows still sum to one.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }


}  // namespace
