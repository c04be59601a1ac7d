// Probe 11: probe 10's ping-pong groups on a RING of four k16 stages (32 KB each: 256 activation rows x 64 B of fp32 +
// 256 weight rows x 64 B of "hl16" planes = per 16 k-values 16 bf16 hi then 16 bf16 lo).  Every LOAD segment issues
// the wave's share (4 LDS-DMA instructions of 16 rows x 64 B) of the stage three k16-steps ahead, so ~3 stages =
// 96 KB per CU are in flight all the time and nobody waits for a DMA that was issued less than five segments ago
// (counted s_waitcnt vmcnt(8): the two newest groups stay outstanding).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p11 scripts/probes/gemm_probe11.hip && /tmp/p11 32768 768 2304
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
// ablations (timing only, results wrong): NO_DMA: no LDS-DMA inside the k loop; NO_SPLIT: fragments reinterpreted instead
// of converted; NO_MFMA: compute segment empty (fragments kept alive); NO_READ: no fragment reads inside the k loop
#ifndef NO_DMA
#define NO_DMA 0
#endif
#ifndef NO_SPLIT
#define NO_SPLIT 0
#endif
#ifndef NO_MFMA
#define NO_MFMA 0
#endif
#ifndef NO_READ
#define NO_READ 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int TM = 256, TN = 256, STAGE = 32768, BOFF = 16384, NSTAGE = 4;

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
  const int ns = K >> 4;                       // k16 steps
  const uint32_t rowb = (uint32_t)K * 4;

  // ---- LDS-DMA: one instruction = 16 rows x 64 B.  Stage image: row pitch 64 B (4 chunks of 16 B), chunk c of row r
  // stored at chunk c ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 lane group touches cover all 16 slots of a bank row.
  // This wave moves row blocks rb = 8g + 2w + j (j = 0, 1) of A and of W.
  const int lrow = lane >> 2, pch = lane & 3;
  const int cl = pch ^ ((lrow >> 2) & 3);      // logical chunk fetched by this lane
  const uint8_t* ga[2];
  const uint8_t* gb[2];
  uint32_t ldst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rb = 8 * g + 2 * w + j;          // 16 row blocks of 16 rows
    int ra = m0 + 16 * rb + lrow; ra = ra < M ? ra : M - 1;
    int rbn = n0 + 16 * rb + lrow; rbn = rbn < Nout ? rbn : Nout - 1;
    ga[j] = A + (size_t)ra * rowb + cl * 16;
    gb[j] = W + (size_t)rbn * rowb + cl * 16;
    ldst[j] = rb * 1024;
  }
  auto issue = [&](int s) {                    // the wave's 4 instructions of k16-step s
    uint8_t* base = lds + (s & (NSTAGE - 1)) * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[j] + (size_t)s * 64),
                                       (lds_void*)(base + ldst[j]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[j] + (size_t)s * 64),
                                       (lds_void*)(base + BOFF + ldst[j]), 16, 0, 0);
    }
  };

  // ---- fragments.  A (fp32): row 64w + 32i + (lane & 31), k = 8lh .. +8 of the step -> logical chunks 2lh, 2lh + 1.
  // W (hl16): row 128g + 32j + (lane & 31), logical chunks hi: lh, lo: 2 + lh.
  const int lr = lane & 31, lh = lane >> 5, swz = (lr >> 2) & 3;
  const uint32_t aoff0 = ((2 * lh) ^ swz) * 16, aoff1 = ((2 * lh + 1) ^ swz) * 16;
  const uint32_t boffh = (lh ^ swz) * 16, boffl = ((2 + lh) ^ swz) * 16;
  const uint32_t arow = (64 * w + lr) * 64, brow = BOFF + (128 * g + lr) * 64;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t ah[2], al[2], bh[4], bl[4];
  auto split8 = [&](const float4 u, const float4 v, bf16x8_t* hi, bf16x8_t* lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    union { bf16x8_t b; uint32_t d[4]; } H, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      union { bf16x2_t b; uint32_t d; } h, l;
      h.b = __builtin_convertvector((f32x2_t){x[2 * p], x[2 * p + 1]}, bf16x2_t);
      const float r0 = x[2 * p] - __uint_as_float(h.d << 16), r1 = x[2 * p + 1] - __uint_as_float(h.d & 0xffff0000u);
      l.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
      H.d[p] = h.d;
      L.d[p] = l.d;
    }
    *hi = H.b;
    *lo = L.b;
  };
  auto load_frags = [&](int s) {
    const uint8_t* st = lds + (s & (NSTAGE - 1)) * STAGE;
    float4 a0[2], a1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      union { bf16x8_t b; float4 f; } u0, u1;   // one LDS access type everywhere (TBAA would otherwise force vmcnt(0))
      u0.b = *reinterpret_cast<const bf16x8_t*>(st + arow + i * 2048 + aoff0);
      u1.b = *reinterpret_cast<const bf16x8_t*>(st + arow + i * 2048 + aoff1);
      a0[i] = u0.f;
      a1[i] = u1.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8_t*>(st + brow + j * 2048 + boffh);
      bl[j] = *reinterpret_cast<const bf16x8_t*>(st + brow + j * 2048 + boffl);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#if NO_SPLIT
      union { float4 f; bf16x8_t b; } x0, x1; x0.f = a0[i]; x1.f = a1[i]; ah[i] = x0.b; al[i] = x1.b;
#else
      split8(a0[i], a1[i], &ah[i], &al[i]);
#endif
    }
  };
  auto compute = [&]() {
#if NO_MFMA
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(ah[i]), "v"(al[i]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(bh[j]), "v"(bl[j]));
    return;
#endif
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  // wait until this wave's DMA group of step s + 3 has landed: the groups of steps s + 4, s + 5 (4 instructions each),
  // as far as they exist, may stay in flight
  auto wait_group = [&](int s) {
    if (s + 5 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (s + 4 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // Segments (s_barrier after each):   group 0: L(0) C(0) L(1) C(1) ...        group 1 one segment behind.
  // L(s) issues step s + 3 into stage (s + 3) % 4 = the stage of step s - 1, whose last reads (group 1's L(s - 1))
  // ended one barrier before group 0's L(s).  Step s + 3 is first read in group 0's L(s + 3); the last barrier before
  // that ends group 0's C(s + 2) and group 1's L(s + 2): that is where the issuing waves wait for the group.
  issue(0);
  if (1 < ns) issue(1);
  if (2 < ns) issue(2);
  if (2 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (1 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();
  for (int s = 0; s < ns; ++s) {
    // LOAD segment
    if (!NO_DMA && s + 3 < ns) issue(s + 3);
    if (!NO_READ || s == 0) load_frags(s);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the group needed next by group 0 is the one of step s + 1: issued at L(s - 2)
    if (g == 1 && !NO_DMA) wait_group(s - 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // COMPUTE segment
    compute();
    if (g == 0 && !NO_DMA) wait_group(s - 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (g == 0) __builtin_amdgcn_s_barrier();

  // epilogue: 32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 64 * w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + 128 * g + j * 32 + lr;
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
      const size_t base = (size_t)r * K * 2 + (size_t)(k >> 4) * 32;
      dst[base + (k & 15)] = h;
      dst[base + 16 + (k & 15)] = l;
    }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, K = argc > 2 ? atoi(argv[2]) : 768, Nout = argc > 3 ? atoi(argv[3]) : 2304;
  std::vector<float> ha((size_t)M * K), hw((size_t)Nout * K);
  srand(1);
  for (auto& x : ha) x = (rand() % 20001 - 10000) / 5000.f;          // full-range random data (DVFS-honest)
  for (auto& x : hw) x = (rand() % 20001 - 10000) / 250000.f;
  std::vector<uint16_t> pw;
  to_hl32(hw, Nout, K, pw);
  uint8_t *A, *W; float* out;
  hipMalloc(&A, ha.size() * 4); hipMalloc(&W, pw.size() * 2); hipMalloc(&out, (size_t)M * Nout * 4);
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, pw.data(), pw.size() * 2, hipMemcpyHostToDevice);
  hipMemset(out, 0, (size_t)M * Nout * 4);
  const int tiles_m = (M + TM - 1) / TM, tiles_n = (Nout + TN - 1) / TN, tiles = tiles_m * tiles_n;
  const size_t ldsb = NSTAGE * STAGE;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(512), ldsb, 0, A, W, out, M, K, Nout, tiles_n, tiles);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(512), ldsb, 0, A, W, out, M, K, Nout, tiles_n, tiles);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("probe11 RING4 %d NO_DMA/SPLIT/MFMA/READ=%d%d%d%d PRIO=%d M=%d K=%d N=%d tiles=%d: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued = %.1f %% of 2500) err=%s\n", WAIT_LATE, NO_DMA, NO_SPLIT, NO_MFMA, NO_READ, PRIO, M, K, Nout,
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
