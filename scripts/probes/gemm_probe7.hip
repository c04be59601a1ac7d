// Probe 7: software-pipelined split GEMM main loop.  Double-buffered LDS (2 x 40 KB -> one 4-wave workgroup per CU),
// ONE barrier per k-tile; the staging of tile t+1 (bf16 split + ds_write into the other buffer) and the global loads of
// tile t+2 are interleaved with the 24 MFMAs of tile t by __builtin_amdgcn_sched_group_barrier.  -DSCHED=0 leaves the
// order to hipcc.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include <math.h>
#ifndef TBK
#define TBK 32
#endif
#ifndef DEPTH
#define DEPTH 2
#endif
#ifndef SCHED
#define SCHED 1
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

__global__ __launch_bounds__(256) void k(const float* A, const uint16_t* W, float* out, int M, int K, int Nout, int tiles_n) {
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
  Regs r0;
  fetch_c(0, r0);
  stage(0, r0);
  fetch_c(1, r0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    compute(cur);
    stage(cur ^ 1, r0);          // tile t+1 (clamped: the last iteration restages tile nk-1 into the idle buffer)
    fetch_c(t + 2, r0);
#if SCHED
    // issue order per wave: 8 fragment reads, then MFMA-paced groups carrying the split VALU, the LDS stores, the second
    // half of the fragment reads and finally the global loads
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
    __syncthreads();
  }
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
  const size_t ldsb = (size_t)2 * 2 * (TBM + TBN) * TSP * 2;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("probe7 SCHED=%d UNR=%d TBK=%d PRIO=%d lds=%zu: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued) err=%s\n", SCHED, UNR, TBK, PRIO, ldsb, us,
         2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6, hipGetErrorString(hipGetLastError()));
  std::vector<float> ho((size_t)M * Nout); hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
  auto bf = [](uint16_t u) { uint32_t x = (uint32_t)u << 16; float f; memcpy(&f, &x, 4); return (double)f; };
  double worst = 0;
  for (int q = 0; q < 64; ++q) {
    const int m = (q * 977) % M, n = (q * 331) % Nout;
    double ref = 0;
    for (int kk = 0; kk < K; ++kk) ref += (double)h[(size_t)m * K + kk] * (bf(hw[(size_t)n * K + kk]) + bf(hw[(size_t)Nout * K + (size_t)n * K + kk]));
    worst = fmax(worst, fabs(ref - ho[(size_t)m * Nout + n]) / (fabs(ref) + 1.0));
  }
  printf("  max rel err over 64 samples: %.2e\n", worst);
  return 0;
}
