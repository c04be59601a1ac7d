// Probe 9: split-precision GEMM on PRE-SPLIT operands ("hl32" planes: per row, per group of 32 k-values,
// 32 bf16 hi then 32 bf16 lo = one 128-byte line), 256x256 workgroup tile, 8 waves as two ping-pong groups
// (group g owns rows [128g, 128g+128), wave w of a group owns columns [64w, 64w+64): 4x2 32x32 accumulators),
// operands streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear destination, XOR swizzle
// applied on the SOURCE chunk), two 64 KB LDS stages.  Per k16-step a group alternates a LOAD segment
// (its quarter of the next k-tile's LDS-DMA + 12 ds_read_b128) with a COMPUTE segment (24 MFMAs) while the
// other group does the opposite, one s_barrier per segment.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p9 scripts/probes/gemm_probe9.hip && /tmp/p9 32768 768 2304
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#ifndef PRIO
#define PRIO 0
#endif
#ifndef WAIT_LATE
#define WAIT_LATE 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int TM = 256, TN = 256, STAGE = 65536, BOFF = 32768;

__global__ __launch_bounds__(512, 2) void k(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W, float* __restrict__ out,
                                            int M, int K, int Nout, int tiles_n, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // XCD-aware tile order: workgroup b runs on XCD b % 8; each XCD walks a contiguous run of row-major tiles
  int tile;
  {
    const int b = blockIdx.x, x = b % 8, s = b / 8, q = tiles_total / 8, r = tiles_total % 8;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
  }
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = wave >> 2, w = wave & 3;
  const int m0 = bm * TM, n0 = bn * TN;
  const int nk = K >> 5;
  const uint32_t rowb = (uint32_t)K * 4;

  // ---- LDS-DMA: this wave's 4 row blocks (8 rows each) of A and of B; lane -> (row in block, physical chunk)
  const int lrow = lane >> 3, pch = lane & 7;
  const uint8_t* ga[4];
  const uint8_t* gb[4];
  uint32_t ldst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rb = 16 * g + 4 * w + j;                       // row block 0..31
    const int c = pch ^ ((lrow >> 1) | ((j & 1) << 2));      // logical chunk stored at this lane's physical slot
    int ra = m0 + 8 * rb + lrow; ra = ra < M ? ra : M - 1;
    int rbn = n0 + 8 * rb + lrow; rbn = rbn < Nout ? rbn : Nout - 1;
    ga[j] = A + (size_t)ra * rowb + c * 16;
    gb[j] = W + (size_t)rbn * rowb + c * 16;
    ldst[j] = rb * 1024;
  }
  auto issue = [&](int t, int half) {  // half 0: A row blocks of this wave, half 1: B row blocks
    uint8_t* base = lds + (t & 1) * STAGE + (half ? BOFF : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint8_t* src = (half ? gb[j] : ga[j]) + (size_t)t * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (lds_void*)(base + ldst[j]), 16, 0, 0);
    }
  };

  // ---- fragment reads: row = base + (lane & 31), logical chunks hi: 2kk + lh, lo: 4 + 2kk + lh
  const int lr = lane & 31, lh = lane >> 5, swz = (lr >> 1) & 7;
  uint32_t coff[2][2];   // [kk][hi/lo] byte offset inside a 128-byte row
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    coff[kk][0] = ((2 * kk + lh) ^ swz) * 16;
    coff[kk][1] = ((4 + 2 * kk + lh) ^ swz) * 16;
  }
  const uint32_t arow = (128 * g + lr) * 128, brow = BOFF + (64 * w + lr) * 128;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t ah[4], al[4], bh[2], bl[2];
  auto load_frags = [&](int t, int kk) {
    const uint8_t* st = lds + (t & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8_t*>(st + arow + i * 4096 + coff[kk][0]);
      al[i] = *reinterpret_cast<const bf16x8_t*>(st + arow + i * 4096 + coff[kk][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8_t*>(st + brow + j * 4096 + coff[kk][0]);
      bl[j] = *reinterpret_cast<const bf16x8_t*>(st + brow + j * 4096 + coff[kk][1]);
    }
  };
  auto compute = [&]() {
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };

  // prologue: k-tile 0, all of it
  issue(0, 0);
  issue(0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one segment behind group 0
  for (int t = 0; t < nk; ++t) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // LOAD segment
      if (t + 1 < nk) issue(t + 1, kk);
      load_frags(t, kk);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // COMPUTE segment
      compute();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
  if (g == 0) __builtin_amdgcn_s_barrier();

  // epilogue: 32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 128 * g + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + 64 * w + j * 32 + lr;
        if (n < Nout) out[(size_t)m * Nout + n] = acc[i][j][r];
      }
    }
}

static uint16_t f2bf(float f) {  // round to nearest even
  uint32_t x; memcpy(&x, &f, 4);
  x += 0x7fff + ((x >> 16) & 1);
  return (uint16_t)(x >> 16);
}
static float bf2f(uint16_t u) { uint32_t x = (uint32_t)u << 16; float f; memcpy(&f, &x, 4); return f; }

// fp32 (R, K) -> hl32 planes
static void to_hl32(const std::vector<float>& src, int R, int K, std::vector<uint16_t>& dst) {
  dst.resize((size_t)R * K * 2);
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < K; ++k) {
      const float x = src[(size_t)r * K + k];
      const uint16_t h = f2bf(x), l = f2bf(x - bf2f(h));
      const size_t base = (size_t)r * K * 2 + (size_t)(k >> 5) * 64;
      dst[base + (k & 31)] = h;
      dst[base + 32 + (k & 31)] = l;
    }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, K = argc > 2 ? atoi(argv[2]) : 768, Nout = argc > 3 ? atoi(argv[3]) : 2304;
  std::vector<float> ha((size_t)M * K), hw((size_t)Nout * K);
  srand(1);
  for (auto& x : ha) x = (rand() % 20001 - 10000) / 5000.f;          // full-range random data (DVFS-honest)
  for (auto& x : hw) x = (rand() % 20001 - 10000) / 250000.f;
  std::vector<uint16_t> pa, pw;
  to_hl32(ha, M, K, pa);
  to_hl32(hw, Nout, K, pw);
  uint8_t *A, *W; float* out;
  hipMalloc(&A, pa.size() * 2); hipMalloc(&W, pw.size() * 2); hipMalloc(&out, (size_t)M * Nout * 4);
  hipMemcpy(A, pa.data(), pa.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, pw.data(), pw.size() * 2, hipMemcpyHostToDevice);
  hipMemset(out, 0, (size_t)M * Nout * 4);
  const int tiles_m = (M + TM - 1) / TM, tiles_n = (Nout + TN - 1) / TN, tiles = tiles_m * tiles_n;
  const size_t ldsb = 2 * STAGE;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(512), ldsb, 0, A, W, out, M, K, Nout, tiles_n, tiles);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(512), ldsb, 0, A, W, out, M, K, Nout, tiles_n, tiles);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("probe9 PRIO=%d M=%d K=%d N=%d tiles=%d: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued = %.1f %% of 2500) err=%s\n", PRIO, M, K, Nout,
         tiles, us, 2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6 / 25.0,
         hipGetErrorString(hipGetLastError()));
  std::vector<float> ho((size_t)M * Nout); hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int q = 0; q < 256; ++q) {
    const int m = (int)(((long long)q * 9773 + 17) % M), n = (int)(((long long)q * 3331 + 5) % Nout);
    double ref = 0;
    for (int kk = 0; kk < K; ++kk) ref += (double)ha[(size_t)m * K + kk] * hw[(size_t)n * K + kk];
    worst = fmax(worst, fabs(ref - ho[(size_t)m * Nout + n]) / (fabs(ref) + 0.05));
  }
  printf("  max rel err over 256 samples: %.2e\n", worst);
  return 0;
}
