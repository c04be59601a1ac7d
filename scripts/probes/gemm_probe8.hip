// Probe 8: split GEMM, 256x256 workgroup tile, 4 waves (one per SIMD), each wave a 128x128 sub-tile = 4x4 MFMA
// accumulators (256 AGPRs).  Per 16-wide k-step a wave reads 16 fragments for 48 MFMAs (0.33 KB of LDS per MFMA, half
// of the 64x64 per-wave tile) and the staging work per MFMA halves too.  Double-buffered LDS (2 x 48 KB), one barrier
// per k-tile, staging of tile t+1 and loads of tile t+2 issued after the MFMAs of tile t in program order (hipcc
// interleaves; -DSCHED=1 adds sched_group_barrier pacing).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#ifndef SCHED
#define SCHED 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int TBM = 256, TBN = 256, TBK = 16, TSP = TBK + 8, NT = 256;
constexpr int PLANE = TBM * TSP;           // elements per plane (TBM == TBN)
constexpr int BUF = 4 * PLANE;             // Ahi, Alo, Bhi, Blo

__device__ __forceinline__ void split4(const float4 v, bf16x4_t* hi, bf16x4_t* lo) {
  const f32x4_t x = {v.x, v.y, v.z, v.w};
  const bf16x4_t h = __builtin_convertvector(x, bf16x4_t);
  const f32x4_t r = x - __builtin_convertvector(h, f32x4_t);
  *hi = h; *lo = __builtin_convertvector(r, bf16x4_t);
}
struct Regs { float4 a[4]; uint4 h[2], l[2]; };

__global__ __launch_bounds__(256, 1) void k(const float* A, const uint16_t* W, float* out, int M, int K, int Nout, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tile = blockIdx.x, bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = bm * TBM, n0 = bn * TBN;
  const int lr = lane & 31, lh = lane >> 5;
  const uint16_t* whi = W; const uint16_t* wlo = W + (int64_t)Nout * K;
  const int nk = K / TBK;
  // A: 4 threads per row (one float4 each), 64 rows per pass, 4 passes.  W: 2 threads per row (8 bf16 each), 128 rows per pass.
  const int ar = tid >> 2, ac = (tid & 3) * 4, wr = tid >> 1, wc = (tid & 1) * 8;
  auto fetch = [&](int t, Regs& r) {
    const int k0 = (t < nk ? t : nk - 1) * TBK;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.a[j] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + ar + 64 * j) * K + k0 + ac);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t o = (int64_t)(n0 + wr + 128 * j) * K + k0 + wc;
      r.h[j] = *reinterpret_cast<const uint4*>(whi + o); r.l[j] = *reinterpret_cast<const uint4*>(wlo + o);
    }
  };
  auto stage = [&](int buf, const Regs& r) {
    __bf16* Ahi = lds + buf * BUF; __bf16* Alo = Ahi + PLANE; __bf16* Bhi = Ahi + 2 * PLANE; __bf16* Blo = Ahi + 3 * PLANE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4_t h, l; split4(r.a[j], &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + (ar + 64 * j) * TSP + ac) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + (ar + 64 * j) * TSP + ac) = l;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<uint4*>(Bhi + (wr + 128 * j) * TSP + wc) = r.h[j];
      *reinterpret_cast<uint4*>(Blo + (wr + 128 * j) * TSP + wc) = r.l[j];
    }
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto compute = [&](int buf) {
    const __bf16* Ahi = lds + buf * BUF; const __bf16* Alo = Ahi + PLANE; const __bf16* Bhi = Ahi + 2 * PLANE; const __bf16* Blo = Ahi + 3 * PLANE;
    bf16x8_t ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ao = (wm * 128 + i * 32 + lr) * TSP + 8 * lh, bo = (wn * 128 + i * 32 + lr) * TSP + 8 * lh;
      ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao); al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
      bh[i] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo); bl[i] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };
  Regs r0;
  fetch(0, r0);
  stage(0, r0);
  fetch(1, r0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    compute(cur);
    stage(cur ^ 1, r0);
    fetch(t + 2, r0);
#if SCHED
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
#pragma unroll
    for (int m = 0; m < 20; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < 4; ++j) out[(int64_t)m * Nout + n0 + wn * 128 + j * 32 + lr] = acc[i][j][r];
    }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, K = argc > 2 ? atoi(argv[2]) : 768, Nout = argc > 3 ? atoi(argv[3]) : 2304;
  float *A, *out; uint16_t* W;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)2 * Nout * K * 2); hipMalloc(&out, (size_t)M * Nout * 4);
  std::vector<float> h((size_t)M * K); for (auto& x : h) x = (rand() % 2000 - 1000) / 1000.f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  // hi plane: random bf16 in [1, 2); lo plane: small (exponent 2^-9), so hi + lo is a plausible split weight
  std::vector<uint16_t> hw((size_t)2 * Nout * K);
  for (size_t i = 0; i < (size_t)Nout * K; ++i) { hw[i] = 0x3f80 + rand() % 128; hw[(size_t)Nout * K + i] = 0x3b00 + rand() % 128; }
  hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const int tiles_n = Nout / TBN, tiles = (M / TBM) * tiles_n;
  const size_t ldsb = (size_t)2 * BUF * 2;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(s);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("probe8 SCHED=%d M=%d K=%d N=%d tiles=%d lds=%zu: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued) err=%s\n", SCHED, M, K, Nout, tiles, ldsb,
         us, 2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6, hipGetErrorString(hipGetLastError()));
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
