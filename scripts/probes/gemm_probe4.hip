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
#ifndef MROWS
#define MROWS 8192
#endif
#ifndef KDIM
#define KDIM 768
#endif
#ifndef NOUT
#define NOUT 2304
#endif
#ifndef WTM
#define WTM 2
#endif
#ifndef WTN
#define WTN 2
#endif
#ifndef WMV
#define WMV 2
#endif
#ifndef WNV
#define WNV 2
#endif
constexpr int WM = WMV, WN = WNV, TBM = WM * WTM * 32, TBN = WN * WTN * 32, NT = WM * WN * 64, TSP = TBK + 8;

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

__global__ __launch_bounds__(NT) void k(const float* A, const uint16_t* W, float* out, int M, int K, int Nout, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* Ahi = lds; __bf16* Alo = lds + TBM * TSP; __bf16* Bhi = lds + 2 * TBM * TSP; __bf16* Blo = Bhi + TBN * TSP;
  const int tile = blockIdx.x, bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int m0 = bm * TBM, n0 = bn * TBN;
  constexpr int ACH = TBK / 4, AROWS = NT / ACH, AJ = TBM / AROWS, WCH = TBK / 8, WROWS = NT / WCH, WJ = TBN / WROWS;
  const int ar0 = tid / ACH, ac4 = tid % ACH, wr0 = tid / WCH, wc8 = (tid % WCH) * 8;
  const uint16_t* whi = W; const uint16_t* wlo = W + (int64_t)Nout * K;
  float4 ra[AJ]; uint4 rwh[WJ], rwl[WJ];
#if ABL & 32
  float4 ra2[AJ]; uint4 rwh2[WJ], rwl2[WJ];
#endif
  for (int j = 0; j < AJ; ++j) ra[j] = make_float4(1, 2, 3, 4);
  for (int j = 0; j < WJ; ++j) { rwh[j] = make_uint4(1, 2, 3, 4); rwl[j] = rwh[j]; }
  auto fetch = [&](int k0, float4* pa, uint4* ph, uint4* pl) {
#if !(ABL & 1)
    for (int j = 0; j < AJ; ++j) pa[j] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + ar0 + AROWS * j) * K + k0 + ac4 * 4);
    for (int j = 0; j < WJ; ++j) {
      const int64_t o = (int64_t)(n0 + wr0 + WROWS * j) * K + k0 + wc8;
      ph[j] = *reinterpret_cast<const uint4*>(whi + o); pl[j] = *reinterpret_cast<const uint4*>(wlo + o);
    }
#endif
  };
  auto stage = [&](float4* pa, uint4* ph, uint4* pl) {
#if !(ABL & 4)
    for (int j = 0; j < AJ; ++j) {
      bf16x4_t h, l; split4(pa[j], &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + (ar0 + AROWS * j) * TSP + ac4 * 4) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + (ar0 + AROWS * j) * TSP + ac4 * 4) = l;
    }
    for (int j = 0; j < WJ; ++j) {
      *reinterpret_cast<uint4*>(Bhi + (wr0 + WROWS * j) * TSP + wc8) = ph[j];
      *reinterpret_cast<uint4*>(Blo + (wr0 + WROWS * j) * TSP + wc8) = pl[j];
    }
#endif
  };
  f32x16 acc[WTM][WTN];
  for (int i = 0; i < WTM; ++i) for (int j = 0; j < WTN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / TBK, lr = lane & 31, lh = lane >> 5;
  bf16x8_t ah[WTM], al[WTM], bh[WTN], bl[WTN];
  for (int i = 0; i < WTM; ++i) { ah[i] = (bf16x8_t)(__bf16)1.0f; al[i] = ah[i]; }
  for (int i = 0; i < WTN; ++i) { bh[i] = (bf16x8_t)(__bf16)1.0f; bl[i] = bh[i]; }
  fetch(0, ra, rwh, rwl);
#if ABL & 32
  fetch(TBK, ra2, rwh2, rwl2);
#endif
  for (int t = 0; t < nk; ++t) {
#if ABL & 32
    if (t & 1) stage(ra2, rwh2, rwl2); else stage(ra, rwh, rwl);
#else
    stage(ra, rwh, rwl);
#endif
    __syncthreads();
#if ABL & 32
    if (t + 2 < nk) { if (t & 1) fetch((t + 2) * TBK, ra2, rwh2, rwl2); else fetch((t + 2) * TBK, ra, rwh, rwl); }
#else
    if (t + 1 < nk) fetch((t + 1) * TBK, ra, rwh, rwl);
#endif
#pragma unroll
    for (int ks = 0; ks < TBK; ks += 16) {
#if !(ABL & 16)
      for (int i = 0; i < WTM; ++i) {
        const int ao = (wm * WTM * 32 + i * 32 + lr) * TSP + ks + 8 * lh;
        ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao); al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
      }
      for (int i = 0; i < WTN; ++i) {
        const int bo = (wn * WTN * 32 + i * 32 + lr) * TSP + ks + 8 * lh;
        bh[i] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo); bl[i] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
      }
#endif
#if !(ABL & 8)
#ifdef REORDER
      for (int i = 0; i < WTM; ++i) for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      for (int i = 0; i < WTM; ++i) for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      for (int i = 0; i < WTM; ++i) for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#else
      for (int i = 0; i < WTM; ++i) for (int j = 0; j < WTN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
#endif
#else
      for (int i = 0; i < WTM; ++i) for (int j = 0; j < WTN; ++j) acc[i][j][0] += (float)ah[i][0] + (float)bl[j][1] + (float)al[i][2] + (float)bh[j][3];
#endif
    }
    __syncthreads();
  }
  for (int i = 0; i < WTM; ++i) for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * WTM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    for (int j = 0; j < WTN; ++j) out[(int64_t)m * Nout + n0 + wn * WTN * 32 + j * 32 + lr] = acc[i][j][r] + ra[0].x + (float)rwh[0].x;
  }
}

int main() {
  const int M = MROWS, K = KDIM, Nout = NOUT;
  float *A, *out; uint16_t* W;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)2 * Nout * K * 2); hipMalloc(&out, (size_t)M * Nout * 4);
  std::vector<float> h((size_t)M * K); for (auto& x : h) x = (rand() % 2000 - 1000) / 1000.f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<uint16_t> hw((size_t)2 * Nout * K); for (auto& x : hw) x = 0x3c00 + rand() % 512;
  hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const int tiles_n = Nout / TBN, tiles = (M / TBM) * tiles_n;
  const size_t LDSB = (size_t)2 * (TBM + TBN) * TSP * 2;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), LDSB, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), LDSB, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("tile %dx%d waves %dx%d ABL=%d TBK=%d: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued)\n", TBM, TBN, WM, WN, ABL, TBK, us, 2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6);
  return 0;
}
