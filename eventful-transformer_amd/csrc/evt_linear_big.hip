// evt_linear_big.hip -- K3/K7, split-precision gated linear for launches that fill the chip: 256-row workgroup tiles.
//
// Why a second kernel.  Ablations of the 128x128 kernel (evt_linear.hip, EVT_ABLATE builds, B = 256 clips) showed that its
// matrix pipe is starved by the operand path, not by instruction issue: MFMAs alone run the K = 3072 product in 240 us, the
// load path alone (global loads + split + LDS staging, no MFMA) takes 299 us, both together 442 us, and dropping only the
// in-loop global loads gives back 25 %.  A 128x128 tile moves 1024 operand bytes through L2 -> CU per k for 32768 FLOP
// (32 FLOP/B); at 340 TF that is ~11 TB/s of L2 reads for A and W tiles, every workgroup waiting ~1.5 us per k-tile for them.
// This kernel halves the bytes per FLOP (256x256: 64 FLOP/B; 256x128: 43) and hides what is left behind a full k-tile:
//
//   * one workgroup per CU, 8 waves (4 x 2): 64x128 per wave (2x4 accumulators of 32x32) for 256x256, 64x64 for 256x128;
//   * TWO LDS stages of hi/lo bf16 tiles (128 KB for 256x256), ONE barrier per k-tile: k-tile t+1 is split and written into
//     the other stage while k-tile t is multiplied; the global loads of k-tile t+2 are in flight meanwhile;
//   * the waves of a SIMD are dealt into two groups that run the two halves of an iteration in opposite order (group 0:
//     stage, then multiply; group 1: multiply, then stage): right after the barrier half of the waves feed the matrix pipe
//     while the other half convert and store, instead of all sixteen converting at once with the pipe idle.
//
// Operands, arithmetic, gather / scatter / p refresh / epilogue are those of gated_linear_split_kernel: results are
// bitwise the same (the k order inside a tile and the accumulation order over tiles do not change).
#include "evt_linear.h"
#include <stdlib.h>
#include <algorithm>

namespace {

#ifndef EVT_EPI_FAST
#define EVT_EPI_FAST 1
#endif
#ifndef EVT_ABLATE   // timing experiments only (results are wrong)
#define EVT_ABLATE 0
#endif

template <int ACT, int TBM, int TBN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 1) void gated_linear_split_big_kernel(const LinArgs g, int tiles_n, int tiles_total) {
  constexpr int NT = WM * WN * 64, TBK = 32;
  constexpr int MI = TBM / WM / 32, NJ = TBN / WN / 32;   // 32x32 accumulators per wave: MI x NJ
  static_assert(TBM == WM * MI * 32 && TBN == WN * NJ * 32, "wave tiles are multiples of 32");
  static_assert(TBM <= NT && TBN <= NT, "one thread per row / column fills the output-row and bias tables");
  // 64-byte LDS rows, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3) (see gated_linear_split_kernel)
  auto lds_off = [](int row, int k) { return row * TBK + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); };
  constexpr int STAGE = 2 * TBM * TBK + 2 * TBN * TBK;   // bf16 elements: A hi, A lo, W hi, W lo
  extern __shared__ __attribute__((aligned(16))) unsigned char evt_gemm_big_smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(evt_gemm_big_smem);
  uint32_t* orow_tab = reinterpret_cast<uint32_t*>(lds + 2 * STAGE);   // 2 x TBM: byte offset of each output row, ~0 = no such row
  float* bias_tab = reinterpret_cast<float*>(orow_tab + 2 * TBM);      // 2 x TBN: bias of the tile's columns

  // Persistent workgroups, XCD-aware (workgroup w runs on XCD w % 8): XCD x owns the contiguous run [run0, run0 + runlen) of
  // row-major tiles and its cx workgroups walk it side by side (workgroup c takes run0 + c, + cx, + 2 cx, ...), so at any
  // time one L2 serves neighbouring column tiles of a few row tiles.
  const int x8 = blockIdx.x % 8, c8 = blockIdx.x / 8;
  const int cx = gridDim.x / 8 + (x8 < (int)(gridDim.x % 8) ? 1 : 0);
  const int q8 = tiles_total / 8, r8 = tiles_total % 8;
  const int run0 = x8 * q8 + min(x8, r8), runlen = q8 + (x8 < r8 ? 1 : 0);
  if (c8 >= runlen) return;
  const int ntile = (runlen - c8 + cx - 1) / cx;
  const int nk = g.K / TBK, total = ntile * nk;   // k-tiles of all my tiles, numbered through

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool stage_first = ((wave >> 2) & 1) == 0;   // waves w, w + 4, w + 8, ... share a SIMD: each SIMD gets both groups
  const int M = g.B * g.kcap;

  constexpr int ACH = TBK / 4, AROWS = NT / ACH, AJ = TBM / AROWS;   // float4 chunks per A row, rows per pass, passes
  // weight tile: TBN rows x one 128-byte hl32 line (4 hi chunks, 4 lo chunks of 16 bytes); chunk id = tid + NT * j
  constexpr int WROWS = NT / 8, WJ = TBN / WROWS;   // rows per pass, passes
  static_assert(AJ >= 1 && WJ >= 1 && WJ <= 4 && TBN % WROWS == 0, "tile / thread-count mismatch");
  const int ar0 = tid / ACH, ac4 = tid % ACH;
  const int wr0 = tid / 8, wpl = (tid >> 2) & 1, wc8 = (tid & 3) * 8;   // row, plane (0 hi, 1 lo), k offset inside the plane
  const int64_t wpitch = hl32_pitch(g.K);
  const bool do_upd = g.p_upd != nullptr;

  // ---- load side: the tile whose k-tiles are being fetched / staged -------------------------------------------------
  // 32-bit byte offsets from the (scalar) base pointers, not 64-bit pointers per row: the kernel sits at the register limit
  // of two waves per SIMD, and past it hipcc starts copying freshly loaded registers (i.e. waits for the prefetch).
  // The launcher guarantees that A and the weight planes are smaller than 4 GB.
  uint32_t a_off[AJ];          // activation rows of this thread (+ its 16-byte column); rows past M read row 0 and are never stored
  int m0_s = 0;                // first row of the tile being staged
  uint32_t w_off[WJ];          // this thread's 16 bytes of the weight rows' hl32 lines (rows clamped past Nout)
  const char* const Abase = reinterpret_cast<const char*>(g.A);
  const char* const Wbase = reinterpret_cast<const char*>(g.Wsplit);
  char* const Pbase = reinterpret_cast<char*>(g.p_upd);
  int bn_s = 0;                // column tile of the tile being staged (p refresh hand-out)
  int nsrc[AJ], nrow = 0;      // next tile: gathered row indices in flight
  float nbias = 0.f;           //            and its bias
  auto tile_of = [&](int seq) { return run0 + c8 + seq * cx; };
  auto issue_indices = [&](int seq) {   // the gate's index lists (and the bias) of tile `seq` -> registers (consumed by enter_tile)
    const int tile_i = tile_of(seq), bm_i = tile_i / tiles_n;
    const int m0 = bm_i * TBM;
    nbias = g.bias[min((tile_i - bm_i * tiles_n) * TBN + (tid < TBN ? tid : 0), g.Nout - 1)];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = m0 + ar0 + AROWS * j;
      nsrc[j] = (g.a_idx != nullptr) ? g.a_idx[m < M ? m : M - 1] : 0;
    }
    const int m = m0 + tid;
    nrow = (g.o_idx != nullptr && tid < TBM) ? g.o_idx[m < M ? m : M - 1] : 0;
  };
  auto enter_tile = [&](int seq) {   // pointers of tile `seq`, its output-row table into half seq & 1
    const int tile = tile_of(seq), bm = tile / tiles_n, bn = tile - bm * tiles_n;
    const int m0 = bm * TBM, n0 = bn * TBN;
    bn_s = bn;
    m0_s = m0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = m0 + ar0 + AROWS * j;
      a_off[j] = ac4 * 16;
      if (m < M) {
        const int b = m / g.kcap, i = m - b * g.kcap;
        a_off[j] = (uint32_t)((b * g.a_rows + ((g.a_idx != nullptr) ? nsrc[j] : i)) * (int)g.lda + ac4 * 4) * 4u;
      }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int n = n0 + wr0 + WROWS * j;
      w_off[j] = (uint32_t)((n < g.Nout ? n : g.Nout - 1) * (int)wpitch + (tid & 7) * 8) * 2u;
    }
    if (tid < TBM) {
      const int m = m0 + tid;
      uint32_t off = ~0u;
      if (m < M) {
        const int b = m / g.kcap, i = m - b * g.kcap;
        off = (uint32_t)((b * g.o_rows + ((g.o_idx != nullptr) ? nrow : i)) * (int)g.ldo) * 4u;
      }
      orow_tab[(seq & 1) * TBM + tid] = off;
    }
    // (the epilogue runs inside the k loop and must not contain a global LOAD: with the bias fetched there hipcc put
    // s_waitcnt vmcnt(0) in front of EVERY multiply, i.e. waited for the k-tile prefetch it had just issued)
    if (tid < TBN) bias_tab[(seq & 1) * TBN + tid] = nbias;
  };
  float4 ra[AJ];
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // (named registers: hipcc left the array form in scratch)
  u32x4_t rw0, rw1, rw2, rw3;
  auto fetch = [&](int kt) {   // k-tile kt of the load-side tile -> registers (K % 32 == 0: whole tiles only)
    const uint32_t kw = (uint32_t)kt * 128u, ka = (uint32_t)kt * (TBK * 4u);
    rw0 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[0] + kw));
    if (WJ > 1) rw1 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[WJ > 1 ? 1 : 0] + kw));
    if (WJ > 2) rw2 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[WJ > 2 ? 2 : 0] + kw));
    if (WJ > 3) rw3 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[WJ > 3 ? 3 : 0] + kw));
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[j] = *reinterpret_cast<const float4*>(Abase + (a_off[j] + ka));
  };
  auto stage = [&](int kt, int s) {   // registers (k-tile kt of the load-side tile) -> LDS stage s; column tile bn refreshes p for kt = bn (mod tiles_n)
    __bf16* Ahi = lds + s * STAGE;
    __bf16* Alo = Ahi + TBM * TBK;
    __bf16* Bhi = Ahi + 2 * TBM * TBK;
    __bf16* Blo = Bhi + TBN * TBK;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      bf16x4_t h, l;
      split4(ra[j], &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + lds_off(ar0 + AROWS * j, ac4 * 4)) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + lds_off(ar0 + AROWS * j, ac4 * 4)) = l;
    }
    __bf16* Bpl = wpl ? Blo : Bhi;
    *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0, wc8)) = rw0;
    if (WJ > 1) *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0 + WROWS, wc8)) = rw1;
    if (WJ > 2) *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0 + 2 * WROWS, wc8)) = rw2;
    if (WJ > 3) *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0 + 3 * WROWS, wc8)) = rw3;
    if (do_upd && (kt % tiles_n) == bn_s) {
#pragma unroll
      for (int j = 0; j < AJ; ++j)
        if (m0_s + ar0 + AROWS * j < M) *reinterpret_cast<float4*>(Pbase + (a_off[j] + (uint32_t)kt * (TBK * 4u))) = ra[j];
    }
  };

  // ---- multiply side ------------------------------------------------------------------------------------------------
  f32x16 acc[MI][NJ];
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  clear();
  const int lr = lane & 31, lh = lane >> 5;
  auto multiply = [&](int s) {
    const __bf16* Ahi = lds + s * STAGE;
    const __bf16* Alo = Ahi + TBM * TBK;
    const __bf16* Bhi = Ahi + 2 * TBM * TBK;
    const __bf16* Blo = Bhi + TBN * TBK;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < TBK; ks += 16) {
      bf16x8_t ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ao = lds_off(wm * (MI * 32) + i * 32 + lr, ks + 8 * lh);
        ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao);
        al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int bo = lds_off(wn * (NJ * 32) + j * 32 + lr, ks + 8 * lh);
        bh[j] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo);
        bl[j] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
      }
#if EVT_ABLATE == 5   // timing experiment: no MFMA, the fragments are only pinned
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
#else
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
#endif
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto epilogue = [&](int seq) {   // bias, activation, scatter of tile `seq` from the accumulators
    const int tile = tile_of(seq), bm = tile / tiles_n, bn = tile - bm * tiles_n;
    const uint32_t* tab = orow_tab + (seq & 1) * TBM + wm * (MI * 32) + 4 * lh;
    const int col0 = bn * TBN + wn * (NJ * 32) + lr;   // this lane's column in accumulator j: col0 + 32 j
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = bias_tab[(seq & 1) * TBN + wn * (NJ * 32) + lr + 32 * j];
    char* const obase = reinterpret_cast<char*>(g.out);
    const uint32_t cb = (uint32_t)col0 * 4u;
    // Interior tile (every row and column exists -- all tiles when M % 256 == 0 and Nout % TBN == 0): straight-line
    // stores, the 16 row offsets of an accumulator row block read as four 16-byte LDS loads.  (The predicated form costs
    // an LDS round trip and five branches per row; the epilogue of one group has to fit in the shadow of the other
    // group's multiply.)
    const bool interior = (bm + 1) * TBM <= M && (bn + 1) * TBN <= g.Nout;
    if (EVT_EPI_FAST && interior) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        uint4 o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = *reinterpret_cast<const uint4*>(tab + i * 32 + 8 * q);   // rows 8 q + 4 lh + 0..3
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint4 o = o4[r >> 2];
          const uint32_t off = ((r & 3) == 0 ? o.x : (r & 3) == 1 ? o.y : (r & 3) == 2 ? o.z : o.w) + cb;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            float v = acc[i][j][r] + bv[j];
            if (ACT == EVT_ACT_GELU_ERF) v = gelu_erf(v);
#if EVT_ABLATE == 6   // timing experiment: no output stores (the arithmetic stays)
            if (v == 12345.678f)
#endif
            *reinterpret_cast<float*>(obase + (off + 128u * j)) = v;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t off = tab[i * 32 + (r & 3) + 8 * (r >> 2)];
        if (off == ~0u) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (col0 + 32 * j < g.Nout) {
            float v = acc[i][j][r] + bv[j];
            if (ACT == EVT_ACT_GELU_ERF) v = gelu_erf(v);
            *reinterpret_cast<float*>(obase + (off + cb + 128u * j)) = v;
          }
        }
      }
    }
  };
  // Raw barrier: __syncthreads() would also drain the global loads and stores that are meant to stay in flight across it.
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // The k-tiles of all my tiles form ONE stream x = 0 .. total - 1 (tile x / nk, k-tile x % nk); LDS stage x & 1.
  // Iteration i stages x = i + 1, fetches x = i + 2, and multiplies x = i (group 0) or x = i + 1 (group 1, which therefore
  // runs "multiply, then stage" between two barriers).  Every wave passes one barrier per iteration: barrier b_i separates all
  // reads of stream element i from the writes of element i + 2 into the same stage, and all writes of i + 1 from its reads.
  // The stream does not stop at a tile boundary: while one group of a SIMD stores a finished tile (epilogue, in front of
  // its first multiply of the next tile), the other group is still multiplying, and the loads of the next tile are already
  // in flight -- no prologue / drain bubble per tile, stores overlap the matrix pipe.
  // (ONE stage site and ONE multiply site in the code: with a copy per group hipcc stops accumulating in place -- twice the
  // accumulator registers.)
  // (ONE fetch site as well: the loop starts two steps early, so that no loaded register is merged with a copy from a prologue
  // fetch -- hipcc resolved that merge by moving freshly loaded registers at the loop's back edge, i.e. by waiting for the loads.)
  issue_indices(0);
  enter_tile(0);
  int sk = 0, sseq = 0;    // load side: k-tile / tile sequence number of the stream element fetched last
  int mk = stage_first ? -2 : -1, mseq = 0;   // multiply side: k-tile / tile of the element this wave multiplies in iteration i
  for (int i = -2; i < total; ++i) {
#if EVT_ABLATE == 2   // timing experiment: staged once
    if (i == -1) stage(sk, 0), stage(sk, 1);
#else
    if (i >= -1 && i + 1 < total) stage(sk, (i + 1) & 1);
#endif
    if (i >= -1 && i + 2 < total) {
      if (sk + 1 == nk) {
        enter_tile(++sseq);
        sk = 0;
      } else {
        ++sk;
        if (sk == nk - 1 && sseq + 1 < ntile) issue_indices(sseq + 1);
      }
    }
#if EVT_ABLATE == 1 || EVT_ABLATE == 2   // timing experiment: fetched once
    if (i == -2)
#endif
    fetch(sk);   // unconditional (the last two iterations re-read the last k-tile): every iteration DEFINES the staging registers

    if (i >= -1 && (!stage_first || i < 0)) barrier();
    if (mk >= 0 && mseq < ntile) {
      if (mk == 0 && mseq > 0) {
        epilogue(mseq - 1);
        clear();
      }
      multiply((mseq * nk + mk) & 1);
    }
    if (++mk == nk) { mk = 0; ++mseq; }
    if (stage_first && i >= 0) barrier();
  }
  epilogue(ntile - 1);
}

int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}

template <int TBM, int TBN, int WM, int WN>
void launch_big_cfg(const LinArgs& a, hipStream_t s) {
  const int M = a.B * a.kcap;
  const int tiles_m = (M + TBM - 1) / TBM, tiles_n = (a.Nout + TBN - 1) / TBN;
  constexpr size_t lds_bytes = (size_t)2 * (2 * TBM * 32 + 2 * TBN * 32) * 2 + (size_t)2 * TBM * 4 + (size_t)2 * TBN * 4;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gated_linear_split_big_kernel<EVT_ACT_GELU_ERF, TBM, TBN, WM, WN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gated_linear_split_big_kernel<EVT_ACT_NONE, TBM, TBN, WM, WN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_set = true;
  }
  const dim3 grid(std::min(tiles_m * tiles_n, cu_count())), block(WM * WN * 64);   // persistent: one workgroup per CU
  if (a.act == EVT_ACT_GELU_ERF)
    hipLaunchKernelGGL((gated_linear_split_big_kernel<EVT_ACT_GELU_ERF, TBM, TBN, WM, WN>), grid, block, lds_bytes, s, a, tiles_n,
                       tiles_m * tiles_n);
  else
    hipLaunchKernelGGL((gated_linear_split_big_kernel<EVT_ACT_NONE, TBM, TBN, WM, WN>), grid, block, lds_bytes, s, a, tiles_n,
                       tiles_m * tiles_n);
}

}  // namespace

// Picks a 256-row tile when the launch has enough of them to fill the chip; returns false when the 128x128 kernel
// (or its split-K form) should run instead.  EVT_GEMM_BIG: 0 never, 1 (default) automatic, 2 always 256x256, 3 always 256x128,
// 4 always 256x192.
bool evt_launch_split_big(const LinArgs& a, hipStream_t s) {
  static const int mode = getenv("EVT_GEMM_BIG") ? atoi(getenv("EVT_GEMM_BIG")) : 1;
  // whole 32-k tiles, at least two of them; top-k gating only (the threshold policy's masked rows stay with the 128x128 kernel,
  // which skips dead tiles)
  if (mode == 0 || a.Wsplit == nullptr || (a.K & 31) != 0 || a.K < 64 || a.count != nullptr) return false;
  // 32-bit byte offsets inside the kernel: activations (and the gate reference, same shape), weight planes and output below 4 GB
  if ((int64_t)a.B * a.a_rows * a.lda * 4 >= ((int64_t)1 << 32) || (int64_t)a.Nout * hl32_pitch(a.K) * 2 >= ((int64_t)1 << 32) ||
      (int64_t)a.B * a.o_rows * a.ldo * 4 >= ((int64_t)1 << 32))
    return false;
  const int M = a.B * a.kcap;
  if (mode == 2) { launch_big_cfg<256, 256, 4, 2>(a, s); return true; }
  if (mode == 3) { launch_big_cfg<256, 128, 4, 2>(a, s); return true; }
  if (mode == 4) { launch_big_cfg<256, 192, 4, 2>(a, s); return true; }
  // One persistent workgroup per CU: a launch of T tiles runs in ceil(T / CUs) rounds.  Take the widest tile whose columns
  // divide Nout (no wasted edge columns) and whose last round is at least 85 % full; measured at M = 32768 (B = 256 clips):
  // 256x256 for Nout = 2304 / 3072 (351 / 483 us vs 380 / 531 for the 128x128 kernel), 256x192 for Nout = 768 (129 / 420 vs
  // 134 / 443 us; 256x256 would leave a quarter of the CUs idle in its second round).
  const int cus = cu_count();
  const int tiles_m = (M + 255) / 256;
  auto fills = [&](int tbn) {
    if (a.Nout % tbn != 0) return false;
    const int tiles = tiles_m * (a.Nout / tbn), rounds = (tiles + cus - 1) / cus;
    return tiles >= cus && tiles * 100 >= rounds * cus * 85;
  };
  if (fills(256)) { launch_big_cfg<256, 256, 4, 2>(a, s); return true; }
  if (fills(192)) { launch_big_cfg<256, 192, 4, 2>(a, s); return true; }
  return false;
}
