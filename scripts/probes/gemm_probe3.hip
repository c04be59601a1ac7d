// Probe 3: split-precision GEMM main loop on LDS-DMA (global_load_lds dwordx4), NS-stage LDS ring, counted
// vmcnt, ONE raw s_barrier per k-tile.  A tile = raw fp32 (split hi/lo when fragments are read), W = bf16
// hi/lo planes.  Source-side XOR swizzle keeps the LDS image lane-linear and the fragment reads low-conflict.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#ifndef NS
#define NS 3
#endif
#ifndef WMV
#define WMV 4
#endif
#ifndef WNV
#define WNV 2
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
constexpr int TBK = 32, WM = WMV, WN = WNV, NW = WM * WN, NT = NW * 64, TBM = WM * 64, TBN = WN * 64;
constexpr int A_BYTES = TBM * TBK * 4, W_BYTES = TBN * TBK * 2, STAGE = A_BYTES + 2 * W_BYTES;
constexpr int A_INST = A_BYTES / 1024, W_INST = W_BYTES / 1024;          // wave-instructions per tile
constexpr int A_PW = A_INST / NW, W_PW = (2 * W_INST) / NW;              // per wave
static_assert(A_INST % NW == 0 && (2 * W_INST) % NW == 0, "tile/wave mismatch");
constexpr int LOADS_PW = A_PW + W_PW;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void glds16(const void* g, char* lds_base_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_ptr_t)lds_base_uniform, 16, 0, 0);
}

__global__ __launch_bounds__(NT) void k(const float* A, const uint16_t* W, float* out, int M, int K, int Nout, int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tile = blockIdx.x, bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, m0 = bm * TBM, n0 = bn * TBN;
  const uint16_t* whi = W; const uint16_t* wlo = W + (int64_t)Nout * K;
  const int nk = K / TBK, lr = lane & 31, lh = lane >> 5;

  // per-lane source pointers (k0 = 0) for this wave's DMA instructions
  const char* asrc[A_PW];
#pragma unroll
  for (int q = 0; q < A_PW; ++q) {
    const int inst = wave * A_PW + q, row = inst * 8 + (lane >> 3), cp = lane & 7, c = cp ^ (row & 7);
    asrc[q] = (const char*)(A + (int64_t)(m0 + row) * K) + c * 16;
  }
  const char* wsrc[W_PW];
  int wdst[W_PW];
#pragma unroll
  for (int q = 0; q < W_PW; ++q) {
    const int inst = wave * W_PW + q, plane = inst / W_INST, pi = inst % W_INST;
    const int row = pi * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
    wsrc[q] = (const char*)((plane ? wlo : whi) + (int64_t)(n0 + row) * K) + c * 16;
    wdst[q] = A_BYTES + plane * W_BYTES + pi * 1024;
  }
  auto issue = [&](int t) {
    char* sb = lds + (t % NS) * STAGE;
    const int64_t ko = (int64_t)t * TBK;
#pragma unroll
    for (int q = 0; q < A_PW; ++q) glds16(asrc[q] + ko * 4, sb + (wave * A_PW + q) * 1024);
#pragma unroll
    for (int q = 0; q < W_PW; ++q) glds16(wsrc[q] + ko * 2, sb + wdst[q]);
  };
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s) if (s < nk) issue(s);
  for (int t = 0; t < nk; ++t) {
    // stage t must have landed: the NS-2 younger stages may stay in flight
    if (t + NS - 2 < nk) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LOADS_PW) : "memory"); }
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < nk) issue(t + NS - 1);   // refills the buffer every wave finished reading before the barrier
    const char* sb = lds + (t % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + lr;
        const int c0 = ks * 4 + 2 * lh;
        const float4 x0 = *reinterpret_cast<const float4*>(sb + row * 128 + ((c0 ^ (row & 7)) * 16));
        const float4 x1 = *reinterpret_cast<const float4*>(sb + row * 128 + (((c0 + 1) ^ (row & 7)) * 16));
        const f32x8_t x = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const bf16x8_t h = __builtin_convertvector(x, bf16x8_t);
        ah[i] = h;
        al[i] = __builtin_convertvector(x - __builtin_convertvector(h, f32x8_t), bf16x8_t);
        const int brow = wn * 64 + i * 32 + lr, c = ks * 2 + lh, pc = (c ^ ((brow >> 2) & 3)) * 16;
        bh[i] = *reinterpret_cast<const bf16x8_t*>(sb + A_BYTES + brow * 64 + pc);
        bl[i] = *reinterpret_cast<const bf16x8_t*>(sb + A_BYTES + W_BYTES + brow * 64 + pc);
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
  }
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    for (int j = 0; j < 2; ++j) out[(int64_t)m * Nout + n0 + wn * 64 + j * 32 + lr] = acc[i][j][r];
  }
}

static uint16_t bf16_rne(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float bf16_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  const int M = 8192, K = 768, Nout = 2304;
  float *A, *out; uint16_t* W;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)2 * Nout * K * 2); hipMalloc(&out, (size_t)M * Nout * 4);
  std::vector<float> h((size_t)M * K), w((size_t)Nout * K);
  for (auto& x : h) x = (rand() % 2000 - 1000) / 1000.f;
  for (auto& x : w) x = (rand() % 2000 - 1000) / 50000.f;
  std::vector<uint16_t> hw((size_t)2 * Nout * K);
  for (size_t i = 0; i < w.size(); ++i) { hw[i] = bf16_rne(w[i]); hw[w.size() + i] = bf16_rne(w[i] - bf16_f(hw[i])); }
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const int tiles_n = Nout / TBN, tiles = (M / TBM) * tiles_n;
  const size_t ldsb = (size_t)NS * STAGE;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipDeviceSynchronize();
  std::vector<float> o((size_t)M * Nout);
  hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int s = 0; s < 400; ++s) {
    const int m = (s * 7919) % M, n = (s * 104729) % Nout;
    double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)h[(size_t)m * K + kk] * w[(size_t)n * K + kk];
    maxerr = fmax(maxerr, fabs(ref - o[(size_t)m * Nout + n])); maxref = fmax(maxref, fabs(ref));
  }
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(s);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), ldsb, 0, A, W, out, M, K, Nout, tiles_n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double us = ms * 1e3 / it;
  printf("probe3 NS=%d tile=%dx%d lds=%zu: %.1f us  %.1f TF algorithmic (%.1f TF bf16 issued)  relerr=%.2e  %s\n", NS, TBM, TBN, ldsb, us,
         2.0 * M * K * Nout / us * 1e-6, 6.0 * M * K * Nout / us * 1e-6, maxerr / maxref, hipGetErrorString(hipGetLastError()));
  return 0;
}
