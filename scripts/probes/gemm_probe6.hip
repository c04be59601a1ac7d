// Probe 2: split GEMM main loop with a 2-deep register prefetch ring + double-buffered LDS, one barrier
// per k-tile (manually unrolled x2 so every register set is statically named).
//   -DDEPTH=1: loads for t+1 issued during MFMA(t) (old structure but 1 barrier); -DDEPTH=2: loads for t+2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef TBK
#define TBK 32
#endif
#ifndef DEPTH
#define DEPTH 2
#endif
#ifndef PD
#define PD 2
#endif
#ifndef UNR
#define UNR 8
#endif
#ifndef PRIO
#define PRIO 1
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int TBM = 128, TBN = 128, WM = 2, WN = 2, NT = 256, TSP = TBK + 8;
constexpr int ACH = TBK / 4, AROWS = NT / ACH, AJ = TBM / AROWS, WCH = TBK / 8, WROWS = NT / WCH, WJ = TBN / WROWS;

__device__ __forceinline__ void split4(const float4 v, bf16x4_t* hi, bf16x4_t* lo) {
  const f32x4_t x = {v.x, v.y, v.z, v.w};
  const bf16x4_t h = __builtin_convertvector(x, bf16x4_t);
  const f32x4_t r = x - __builtin_convertvector(h, f32x4_t);
  *hi = h; *lo = __builtin_convertvector(r, bf16x4_t);
}
struct Regs { float4 a[AJ]; uint4 h[WJ], l[WJ]; };

#ifndef MINB
#define MINB 1
#endif
__global__ __launch_bounds__(256, MINB) void k(const float* A, const uint16_t* W, float* out, int M, int K, int Nout, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  constexpr int BUF = 2 * (TBM + TBN) * TSP;
  const int tile = blockIdx.x, bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int m0 = bm * TBM, n0 = bn * TBN;
  const int ar0 = tid / ACH, ac4 = tid % ACH, wr0 = tid / WCH, wc8 = (tid % WCH) * 8;
  const uint16_t* whi = W; const uint16_t* wlo = W + (int64_t)Nout * K;
  const int nk = K / TBK, lr = lane & 31, lh = lane >> 5;
  auto fetch = [&](int t, Regs& r) {
    if (t >= nk) return;
    const int k0 = t * TBK;
#pragma unroll
    for (int j = 0; j < AJ; ++j) r.a[j] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + ar0 + AROWS * j) * K + k0 + ac4 * 4);
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int64_t o = (int64_t)(n0 + wr0 + WROWS * j) * K + k0 + wc8;
      r.h[j] = *reinterpret_cast<const uint4*>(whi + o); r.l[j] = *reinterpret_cast<const uint4*>(wlo + o);
    }
  };
  auto fetch_c = [&](int t, Regs& r) {
    const int k0 = (t < nk ? t : nk - 1) * TBK;
#pragma unroll
    for (int j = 0; j < AJ; ++j) r.a[j] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + ar0 + AROWS * j) * K + k0 + ac4 * 4);
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int64_t o = (int64_t)(n0 + wr0 + WROWS * j) * K + k0 + wc8;
      r.h[j] = *reinterpret_cast<const uint4*>(whi + o); r.l[j] = *reinterpret_cast<const uint4*>(wlo + o);
    }
  };
  auto stage = [&](int buf, const Regs& r) {
    __bf16* Ahi = lds + buf * BUF; __bf16* Alo = Ahi + TBM * TSP; __bf16* Bhi = Ahi + 2 * TBM * TSP; __bf16* Blo = Bhi + TBN * TSP;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      bf16x4_t h, l; split4(r.a[j], &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + (ar0 + AROWS * j) * TSP + ac4 * 4) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + (ar0 + AROWS * j) * TSP + ac4 * 4) = l;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      *reinterpret_cast<uint4*>(Bhi + (wr0 + WROWS * j) * TSP + wc8) = r.h[j];
      *reinterpret_cast<uint4*>(Blo + (wr0 + WROWS * j) * TSP + wc8) = r.l[j];
    }
  };
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto compute = [&](int buf) {
    const __bf16* Ahi = lds + buf * BUF; const __bf16* Alo = Ahi + TBM * TSP; const __bf16* Bhi = Ahi + 2 * TBM * TSP; const __bf16* Blo = Bhi + TBN * TSP;
#if PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ks = 0; ks < TBK; ks += 16) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ao = (wm * 64 + i * 32 + lr) * TSP + ks + 8 * lh, bo = (wn * 64 + i * 32 + lr) * TSP + ks + 8 * lh;
        ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao); al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
        bh[i] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo); bl[i] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
#if PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // single LDS buffer (2 barriers per k-tile, 2 workgroups/CU as in the product kernel), register prefetch distance PD:
  // PD=1: loads for t+1 issued after staging t (product kernel today); PD=2: two named register sets, loads for t+2.
  // Loads are UNCONDITIONAL (tile index clamped) so the loop body is straight-line and hipcc can count vmcnt.
  Regs r0, r1;
#if PD == 2
  // hipcc's waitcnt pass merges the back-edge conservatively: at the loop header it waits vmcnt(0), which collapses a
  // 2-deep ring to depth ~1.  Unrolling the body UNR/2 times keeps UNR-2 of every UNR waits exact (vmcnt(8)).
  fetch_c(0, r0); fetch_c(1, r1);
  for (int t = 0; t < nk; t += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; u += 2) {
      stage(0, r0);
      __syncthreads();
      fetch_c(t + u + 2, r0);
      compute(0);
      __syncthreads();
      stage(0, r1);
      __syncthreads();
      fetch_c(t + u + 3, r1);
      compute(0);
      __syncthreads();
    }
  }
#else
  fetch_c(0, r0);
  for (int t = 0; t < nk; ++t) {
    stage(0, r0);
    __syncthreads();
    fetch_c(t + 1, r0);
    compute(0);
    __syncthreads();
  }
#endif
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    for (int j = 0; j < 2; ++j) out[(int64_t)m * Nout + n0 + wn * 64 + j * 32 + lr] = acc[i][j][r];
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
  const size_t ldsb = (size_t)2 * (TBM + TBN) * TSP * 2;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("probe6 PD=%d UNR=%d TBK=%d PRIO=%d lds=%zu: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued) err=%s\n", PD, UNR, TBK, PRIO, ldsb, us,
         2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6, hipGetErrorString(hipGetLastError()));
  return 0;
}
