// Standalone ablation probe for the split-precision GEMM main loop (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -DABL=<mask> gemm_probe.hip -o probe && ./probe
//   ABL bit0: no global loads (registers reused)   bit1: no split VALU (reinterpret)   bit2: no LDS stage writes
//   bit3: no MFMA                                   bit4: no fragment LDS reads        bit5: prefetch distance 2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef ABL
#define ABL 0
#endif
#ifndef TBK
#define TBK 32
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int TBM = 128, TBN = 128, WM = 2, WN = 2, NT = 256, TSP = TBK + 8;
constexpr int W_BYTES = TBN * TBK * 2;   // one plane, unpadded 64-byte rows (DMA image is lane-linear)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void glds16(const void* g, char* l) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_ptr_t)l, 16, 0, 0); }

__device__ __forceinline__ void split4(const float4 v, bf16x4_t* hi, bf16x4_t* lo) {
#if ABL & 2
  union { float4 f; bf16x4_t b[2]; } u; u.f = v; *hi = u.b[0]; *lo = u.b[1];
#else
  const f32x4_t x = {v.x, v.y, v.z, v.w};
  const bf16x4_t h = __builtin_convertvector(x, bf16x4_t);
  const f32x4_t r = x - __builtin_convertvector(h, f32x4_t);
  *hi = h; *lo = __builtin_convertvector(r, bf16x4_t);
#endif
}

#ifndef MINB
#define MINB 1
#endif
__global__ __launch_bounds__(256, MINB) void k(const float* A, const uint16_t* W, float* out, int M, int K, int Nout, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) char ldsw[2][2 * W_BYTES];   // [buffer][hi plane | lo plane]
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * TBM * TSP];
  __bf16* Ahi = lds; __bf16* Alo = lds + TBM * TSP;
  const int tile = blockIdx.x, bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int m0 = bm * TBM, n0 = bn * TBN;
  constexpr int ACH = TBK / 4, AROWS = NT / ACH, AJ = TBM / AROWS, WCH = TBK / 8, WROWS = NT / WCH, WJ = TBN / WROWS;
  const int ar0 = tid / ACH, ac4 = tid % ACH, wr0 = tid / WCH, wc8 = (tid % WCH) * 8;
  const uint16_t* whi = W; const uint16_t* wlo = W + (int64_t)Nout * K;
  float4 ra[AJ]; uint4 rwh[WJ], rwl[WJ];
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const char* wsrc[4]; int wdst[4];
  for (int q = 0; q < 4; ++q) { const int inst = wv * 4 + q, plane = inst >> 3, pi = inst & 7; const int row = pi * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
    wsrc[q] = (const char*)((plane ? wlo : whi) + (int64_t)(n0 + row) * K) + c * 16; wdst[q] = plane * W_BYTES + pi * 1024; }
  auto dma_w = [&](int t) { char* base = ldsw[t & 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(wsrc[q] + (int64_t)t * TBK * 2, base + wdst[q]); };
#if ABL & 32
  float4 ra2[AJ]; uint4 rwh2[WJ], rwl2[WJ];
#endif
  for (int j = 0; j < AJ; ++j) ra[j] = make_float4(1, 2, 3, 4);
  for (int j = 0; j < WJ; ++j) { rwh[j] = make_uint4(1, 2, 3, 4); rwl[j] = rwh[j]; }
  auto fetch = [&](int k0, float4* pa, uint4* ph, uint4* pl) {
#if !(ABL & 1)
    for (int j = 0; j < AJ; ++j) pa[j] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + ar0 + AROWS * j) * K + k0 + ac4 * 4);
#endif
  };
  auto stage = [&](float4* pa, uint4* ph, uint4* pl) {
#if !(ABL & 4)
    for (int j = 0; j < AJ; ++j) {
      bf16x4_t h, l; split4(pa[j], &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + (ar0 + AROWS * j) * TSP + ac4 * 4) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + (ar0 + AROWS * j) * TSP + ac4 * 4) = l;
    }
#endif
  };
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / TBK, lr = lane & 31, lh = lane >> 5;
  bf16x8_t ah[2], al[2], bh[2], bl[2];
  for (int i = 0; i < 2; ++i) { ah[i] = (bf16x8_t)(__bf16)1.0f; al[i] = ah[i]; bh[i] = ah[i]; bl[i] = ah[i]; }
  fetch(0, ra, rwh, rwl);
  dma_w(0);
#if ABL & 32
  fetch(TBK, ra2, rwh2, rwl2);
#endif
  for (int t = 0; t < nk; ++t) {
#if ABL & 32
    if (t & 1) stage(ra2, rwh2, rwl2); else stage(ra, rwh, rwl);
#else
    stage(ra, rwh, rwl);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if ABL & 32
    if (t + 2 < nk) { if (t & 1) fetch((t + 2) * TBK, ra2, rwh2, rwl2); else fetch((t + 2) * TBK, ra, rwh, rwl); }
#else
    if (t + 1 < nk) { fetch((t + 1) * TBK, ra, rwh, rwl); dma_w(t + 1); }
#endif
#pragma unroll
    for (int ks = 0; ks < TBK; ks += 16) {
#if !(ABL & 16)
      for (int i = 0; i < 2; ++i) {
        const int ao = (wm * 64 + i * 32 + lr) * TSP + ks + 8 * lh;
        ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao); al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
        const int brow = wn * 64 + i * 32 + lr, pc = (((ks >> 3) + lh) ^ ((brow >> 2) & 3)) * 16;
        bh[i] = *reinterpret_cast<const bf16x8_t*>(ldsw[t & 1] + brow * 64 + pc);
        bl[i] = *reinterpret_cast<const bf16x8_t*>(ldsw[t & 1] + W_BYTES + brow * 64 + pc);
      }
#endif
#if !(ABL & 8)
#ifdef REORDER
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#else
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
#endif
#else
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)ah[i][0] + (float)bl[j][1] + (float)al[i][2] + (float)bh[j][3];
#endif
    }
    __syncthreads();
  }
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    for (int j = 0; j < 2; ++j) out[(int64_t)m * Nout + n0 + wn * 64 + j * 32 + lr] = acc[i][j][r] + ra[0].x + (float)rwh[0].x;
  }
}

int main() {
  const int M = 8192, K = 768, Nout = 2304;
  float *A, *out; uint16_t* W;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)2 * Nout * K * 2); hipMalloc(&out, (size_t)M * Nout * 4);
  std::vector<float> h((size_t)M * K); for (auto& x : h) x = (rand() % 2000 - 1000) / 1000.f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<uint16_t> hw((size_t)2 * Nout * K); for (auto& x : hw) x = 0x3c00 + rand() % 512;
  hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const int tiles_n = Nout / TBN, tiles = (M / TBM) * tiles_n;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 0, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 0, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("ABL=%d TBK=%d: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued)\n", ABL, TBK, us, 2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6);
  return 0;
}
