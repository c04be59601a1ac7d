// evt_linear_big.hip -- K3/K7, split-precision gated linear for launches that fill the chip: 256-row workgroup tiles.
//
// Why a second kernel.  Ablations of the 128x128 kernel (evt_linear.hip, EVT_ABLATE builds, B = 256 clips): MFMAs alone run
// the K = 3072 product in 240 us, the load path alone (global loads + split + LDS staging, no MFMA) takes 299 us, both together
// 442 us, and dropping only the in-loop global loads gives back 25 %.  A 128x128 tile moves 1024 operand bytes through
// L2 -> CU per k for 32768 FLOP (32 FLOP/B): ~11 TB/s of L2 reads at 340 TF.  This kernel halves the bytes per FLOP (256x256:
// 64 FLOP/B; 256x192: 55) and the per-element staging work (each activation is converted for half as many column tiles):
//
//   * ONE persistent workgroup per CU, 8 waves (4 x 2): 64x128 per wave (2x4 accumulators of 32x32) for 256x256, 64x96 for
//     256x192; the workgroups of an XCD walk a contiguous run of tiles side by side;
//   * TWO LDS stages of hi/lo bf16 tiles (128 KB for 256x256), ONE barrier per k-tile: k-tile t+1 is split and written into
//     the other stage while k-tile t is multiplied; the global loads of k-tile t+2 are in flight meanwhile; the k-tile stream
//     does not stop at a tile boundary (the next tile's loads are in flight while a finished tile is stored);
//   * the waves of a SIMD are dealt into two groups that run the two halves of an iteration in opposite order (group 0:
//     stage, then multiply; group 1: multiply, then stage).  With fp32 activations this measured neutral (the staging segment
//     is longer than a multiply); with pre-split activations it becomes a two-barrier ping-pong (PP below).
//
// What the in-kernel phase profile (-DEVT_PROF, scripts/gemm_prof.py) says is left: the loads are not waited for; the kernel
// is issue-bound (~250 non-MFMA instructions per wave and k-tile beside 36-48 MFMAs).  DESIGN.md section 6.
//
// Operands, arithmetic, gather / scatter / p refresh / epilogue are those of gated_linear_split_kernel: results are
// bitwise the same (the k order inside a tile and the accumulation order over tiles do not change).
#include "evt_linear.h"
#include <stdlib.h>
#include <algorithm>

namespace {

#ifndef EVT_BIG_PRIO   // 1: multiply segments at priority 1 (default); 0: no priority changes; 2: everything BUT the multiply at priority 1
#define EVT_BIG_PRIO 1
#endif
#ifndef EVT_PINGPONG
#define EVT_PINGPONG 1
#endif
#ifndef EVT_EPI_FAST
#define EVT_EPI_FAST 1
#endif
#ifdef EVT_PROF   // phase timing of two waves of workgroup 0 (scripts/gemm_prof.py): s_memtime at the phase boundaries
__device__ unsigned long long evt_prof_buf[2][8];
#define EVT_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define EVT_TICK(slot) do { } while (0)
#endif
#ifndef EVT_ABLATE   // timing experiments only (results are wrong)
#define EVT_ABLATE 0
#endif

// FMT bit 2 (ABF): the activations are ONE bf16 plane (row pitch lda elements): values that are exactly representable in
// bf16 -- the A.v state of a bf16 `matmul_2_cast`, which IS the attention output (blocks.py:183-189) -- have hi = the value and
// lo = 0, so the tile is staged as a plain copy of half the bytes and the A_lo . W_hi MFMA is skipped (adding its exact zeros
// changes nothing: results are bit-identical to the fp32-activation launch).  The gate reference is refreshed with the widened
// values.
// FMT bit 0 (APL): the activations are already split -- A holds hl32 lines like the weights (row pitch lda * 4 bytes, i.e. the
// bytes of the fp32 row it replaces); staging is then a plain 16-byte copy, no conversion.  FMT bit 1 (OPL): the output is
// written as hl32 lines instead of fp32 (same bytes).  evt_gated_mlp uses both for its hidden scratch: the first launch's
// epilogue splits GELU(x) once per element instead of the second launch splitting it once per column tile in its k loop.
template <int ACT, int TBM, int TBN, int WM, int WN, int DEPTH, int FMT>
__global__ __launch_bounds__(WM * WN * 64, 1) void gated_linear_split_big_kernel(const LinArgs g, int tiles_n, int tiles_total) {
  constexpr int NT = WM * WN * 64, TBK = 32;
  constexpr bool APL = (FMT & 1) != 0, OPL = (FMT & 2) != 0, ABF = (FMT & 4) != 0;
  static_assert(!(APL && ABF), "one activation format");
  constexpr int MI = TBM / WM / 32, NJ = TBN / WN / 32;   // 32x32 accumulators per wave: MI x NJ
  // Ping-pong pacing (pre-split activations only): TWO barriers per k-tile, the multiply-first group running one barrier
  // interval behind, so that on every SIMD one wave multiplies ALONE (all of its fragment reads requested up front) while its
  // partner stages and fetches.  In-kernel phase timing (scripts/gemm_prof.py): with one barrier the two multiplies of a SIMD
  // overlap, the younger wave gets the matrix pipe only in the older one's LDS waits, and its staging then runs with nothing
  // beside it (4000 ticks per k-tile; 36 MFMAs = 1152 per wave).  With fp32 activations the staging segment (48 conversion
  // VALU per k-tile, starved by the partner's MFMA priority: ~2200 ticks) is longer than a multiply and ping-pong loses.
  constexpr bool PP = EVT_PINGPONG != 0 && APL && MI * NJ <= 6;   // (the 2x4 wave tile has no registers for a second fragment set)
  static_assert(TBM == WM * MI * 32 && TBN == WN * NJ * 32, "wave tiles are multiples of 32");
  static_assert(TBM <= NT && TBN <= NT, "one thread per row / column fills the output-row and bias tables");
  // 64-byte LDS rows, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3) (see gated_linear_split_kernel)
  auto lds_off = [](int row, int k) { return row * TBK + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); };
  // The weight lo plane starts 64 bytes (16 banks) past a multiple of 128: the 8 lanes of a ds_write_b128 group store the 4 hi and
  // the 4 lo chunks of one weight row (one 128-byte hl32 line of global memory), which must not meet in the same banks.
  constexpr int WPAD = 32, APAD = APL ? 32 : 0;   // (pre-split activations are staged the same way)
  constexpr int STAGE = 2 * TBM * TBK + APAD + 2 * TBN * TBK + WPAD;   // bf16 elements: A hi, (pad), A lo, W hi, (pad), W lo
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
#ifdef EVT_NO_STAGGER   // timing experiment
  const bool stage_first = true;
#else
  const bool stage_first = ((wave >> 2) & 1) == 0;
#endif   // waves w, w + 4, w + 8, ... share a SIMD: each SIMD gets both groups
  const int M = g.B * g.kcap;
#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == 8 && (wave == 0 || wave == 4);
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif

  constexpr int ACH = ABF ? TBK / 8 : TBK / 4, AROWS = NT / ACH, AJ = TBM / AROWS;   // 16-byte chunks per A row (fp32 / hl32: 8, bf16: 4), rows per pass, passes
  constexpr uint32_t ABYTES = ABF ? TBK * 2u : TBK * 4u;   // bytes of one k-tile of an activation row
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
  auto tile_of = [&](int seq) __attribute__((always_inline)) { return run0 + c8 + seq * cx; };
  auto issue_indices = [&](int seq) __attribute__((always_inline)) {   // the gate's index lists (and the bias) of tile `seq` -> registers (consumed by enter_tile)
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
  auto enter_tile = [&](int seq) __attribute__((always_inline)) {   // pointers of tile `seq`, its output-row table into half seq & 1
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
        a_off[j] = (uint32_t)((b * g.a_rows + ((g.a_idx != nullptr) ? nsrc[j] : i)) * (int)g.lda) * (ABF ? 2u : 4u) + ac4 * 16u;
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
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  // Two sets of staging registers (stream elements of even / odd parity): the loads of element x + 2 are issued as soon as
  // element x has been written to LDS, so a load has TWO iterations to arrive.  (With one set the loop could not turn
  // faster than one load latency + one staging per iteration: ~1.4 us, more than the iteration's MFMA time.)
  struct Regs {   // (named weight registers: hipcc left an array of them in scratch)
    f32x4 a[AJ];   // (native vector type: arrays of HIP's float4 / uint4 structs were left in scratch)
    u32x4_t w0, w1, w2, w3;
  };
  Regs R0, R1;
  auto fetch = [&](Regs& R, int kt) __attribute__((always_inline)) {   // k-tile kt of the load-side tile -> registers (K % 32 == 0: whole tiles only)
    const uint32_t kw = (uint32_t)kt * 128u, ka = (uint32_t)kt * ABYTES;
    R.w0 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[0] + kw));
    if (WJ > 1) R.w1 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[WJ > 1 ? 1 : 0] + kw));
    if (WJ > 2) R.w2 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[WJ > 2 ? 2 : 0] + kw));
    if (WJ > 3) R.w3 = *reinterpret_cast<const u32x4_t*>(Wbase + (w_off[WJ > 3 ? 3 : 0] + kw));
#pragma unroll
    for (int j = 0; j < AJ; ++j) R.a[j] = *reinterpret_cast<const f32x4*>(Abase + (a_off[j] + ka));
  };
  // stage side: the element written to LDS lags the load side by two elements, so it keeps its own tile description
  uint32_t st_off[AJ];
  int st_k = 0, st_bn = 0, st_m0 = 0, st_upd = 0;   // st_upd: next k-tile whose rows this column tile writes back to the gate reference
  auto stage = [&](const Regs& R, int s) __attribute__((always_inline)) {   // registers -> LDS stage s; column tile bn refreshes p for k-tile = bn (mod tiles_n)
#ifdef EVT_PROF
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH == 2 ? AJ + WJ : 0));
    EVT_TICK(6);   // wait for the loads
#endif
    if (st_k == nk) {   // first k-tile of the next tile: the load side entered it two elements ago (and no later tile yet)
      st_k = 0;
      st_bn = bn_s;
      st_upd = bn_s;
      st_m0 = m0_s;
#pragma unroll
      for (int j = 0; j < AJ; ++j) st_off[j] = a_off[j];
    }
    __bf16* Ahi = lds + s * STAGE;
    __bf16* Alo = Ahi + TBM * TBK + APAD;
    __bf16* Bhi = Alo + TBM * TBK;
    __bf16* Blo = Bhi + TBN * TBK + WPAD;
    if (ABF) {   // 8 bf16 values = the hi plane's chunk ac4; there is no lo plane
#pragma unroll
      for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(Ahi + lds_off(ar0 + AROWS * j, ac4 * 8)) = R.a[j];
    } else if (APL) {   // the 16 bytes are chunk ac4 of the row's 128-byte line: chunks 0-3 hi, 4-7 lo
      __bf16* Apl = (ac4 & 4) ? Alo : Ahi;
#pragma unroll
      for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(Apl + lds_off(ar0 + AROWS * j, (ac4 & 3) * 8)) = R.a[j];
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        bf16x4_t h, l;
        split4(make_float4(R.a[j].x, R.a[j].y, R.a[j].z, R.a[j].w), &h, &l);
        *reinterpret_cast<bf16x4_t*>(Ahi + lds_off(ar0 + AROWS * j, ac4 * 4)) = h;
        *reinterpret_cast<bf16x4_t*>(Alo + lds_off(ar0 + AROWS * j, ac4 * 4)) = l;
      }
    }
    __bf16* Bpl = wpl ? Blo : Bhi;
    *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0, wc8)) = R.w0;
    if (WJ > 1) *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0 + WROWS, wc8)) = R.w1;
    if (WJ > 2) *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0 + 2 * WROWS, wc8)) = R.w2;
    if (WJ > 3) *reinterpret_cast<u32x4_t*>(Bpl + lds_off(wr0 + 3 * WROWS, wc8)) = R.w3;
    if (!APL && do_upd && st_k == st_upd) {   // k-tiles st_bn, st_bn + tiles_n, ... (a counter: the modulo cost 14 scalar instructions per k-tile)
      st_upd += tiles_n;
#pragma unroll
      for (int j = 0; j < AJ; ++j)
        if (st_m0 + ar0 + AROWS * j < M) {
          if (ABF) {   // widen the 8 bf16 values: the fp32 reference row has twice the byte offset
            union { f32x4 v; uint32_t u[4]; } in;
            in.v = R.a[j];
            f32x4 lo4, hi4;
            lo4.x = __uint_as_float(in.u[0] << 16); lo4.y = __uint_as_float(in.u[0] & 0xffff0000u);
            lo4.z = __uint_as_float(in.u[1] << 16); lo4.w = __uint_as_float(in.u[1] & 0xffff0000u);
            hi4.x = __uint_as_float(in.u[2] << 16); hi4.y = __uint_as_float(in.u[2] & 0xffff0000u);
            hi4.z = __uint_as_float(in.u[3] << 16); hi4.w = __uint_as_float(in.u[3] & 0xffff0000u);
            char* dst = Pbase + (2u * st_off[j] + (uint32_t)st_k * (TBK * 4u));
            *reinterpret_cast<f32x4*>(dst) = lo4;
            *reinterpret_cast<f32x4*>(dst + 16) = hi4;
          } else {
            *reinterpret_cast<f32x4*>(Pbase + (st_off[j] + (uint32_t)st_k * (TBK * 4u))) = R.a[j];
          }
        }
    }
    ++st_k;
  };

  // ---- multiply side ------------------------------------------------------------------------------------------------
  f32x16 acc[MI][NJ];
  auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  clear();
  const int lr = lane & 31, lh = lane >> 5;
  auto multiply = [&](int s) __attribute__((always_inline)) {
    const __bf16* Ahi = lds + s * STAGE;
    const __bf16* Alo = Ahi + TBM * TBK + APAD;
    const __bf16* Bhi = Alo + TBM * TBK;
    const __bf16* Blo = Bhi + TBN * TBK + WPAD;
    if (EVT_BIG_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (EVT_BIG_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    constexpr int HB = PP ? 2 : 1;   // k halves whose fragments are requested together
#pragma unroll
    for (int ks0 = 0; ks0 < TBK; ks0 += 16 * HB) {
      // (without ping-pong, requesting both halves up front -- 20 reads per wave, 40 more registers -- measured 4 % slower)
      bf16x8_t ah[HB][MI], al[HB][MI], bh[HB][NJ], bl[HB][NJ];
#pragma unroll
      for (int h = 0; h < HB; ++h) {
        const int ks = ks0 + 16 * h;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int ao = lds_off(wm * (MI * 32) + i * 32 + lr, ks + 8 * lh);
          ah[h][i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao);
          if (!ABF) al[h][i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int bo = lds_off(wn * (NJ * 32) + j * 32 + lr, ks + 8 * lh);
          bh[h][j] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo);
          bl[h][j] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
        }
      }
      if (PP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < HB; ++h) {
#if EVT_ABLATE == 5   // timing experiment: no MFMA, the fragments are only pinned
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(ah[h][i]), "v"(al[h][i]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bh[h][j]), "v"(bl[h][j]));
#else
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (!ABF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[h][i], bh[h][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[h][i], bl[h][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[h][i], bh[h][j], acc[i][j], 0, 0, 0);
          }
#endif
      }
    }
    if (EVT_BIG_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (EVT_BIG_PRIO == 2) __builtin_amdgcn_s_setprio(1);
  };
  auto epilogue = [&](int seq) __attribute__((always_inline)) {   // bias, activation, scatter of tile `seq` from the accumulators
    const int tile = tile_of(seq), bm = tile / tiles_n, bn = tile - bm * tiles_n;
    const uint32_t* tab = orow_tab + (seq & 1) * TBM + wm * (MI * 32) + 4 * lh;
    const int col0 = bn * TBN + wn * (NJ * 32) + lr;   // this lane's column in accumulator j: col0 + 32 j
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = bias_tab[(seq & 1) * TBN + wn * (NJ * 32) + lr + 32 * j];
    char* const obase = reinterpret_cast<char*>(g.out);
    // fp32 output: this lane's column.  hl32 output (OPL): the 32 lanes of an accumulator row hold the 32 k of one 128-byte
    // line; neighbouring lanes swap (DPP) so that even lanes store two hi values and odd lanes two lo values as one dword.
    const uint32_t cb = OPL ? (uint32_t)(bn * TBN + wn * (NJ * 32)) * 4u + (uint32_t)((lr & 1) * 64 + (lr >> 1) * 4) : (uint32_t)col0 * 4u;
    auto finish = [&](float v) __attribute__((always_inline)) {   // activation; OPL: the dword this lane stores
      if (ACT == EVT_ACT_GELU_ERF) v = gelu_erf(v);
      if (!OPL) return v;
      asm volatile("" : "+v"(v));   // v is the ROUNDED fp32 value: no contraction of its last multiply into the residual below
      union { bf16x2_t b; uint32_t u; } h, l;
      h.b = __builtin_convertvector((f32x2_t){v, v}, bf16x2_t);
      const float r = v - __uint_as_float(h.u << 16);
      l.b = __builtin_convertvector((f32x2_t){r, r}, bf16x2_t);
      const uint32_t mine = (h.u & 0xffffu) | (l.u << 16);   // hi | lo << 16
      const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      const uint32_t word = (lr & 1) ? ((other >> 16) | (mine & 0xffff0000u))      // lo(k - 1), lo(k)
                                     : ((mine & 0xffffu) | (other << 16));         // hi(k), hi(k + 1)
      return __uint_as_float(word);
    };
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
        if (OPL) {   // two rows at a time: packed conversions, byte permutes instead of shifts and masks
          const uint32_t sel_lane = (lr & 1) ? 0x03020706u : 0x05040100u;   // odd: lo(k-1), lo(k); even: hi(k), hi(k+1)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const uint4 o = o4[r >> 2];
            const uint32_t off0 = ((r & 3) == 0 ? o.x : o.z) + cb, off1 = ((r & 3) == 0 ? o.y : o.w) + cb;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              float v0 = acc[i][j][r] + bv[j], v1 = acc[i][j][r + 1] + bv[j];
              if (ACT == EVT_ACT_GELU_ERF) {
                v0 = gelu_erf(v0);
                v1 = gelu_erf(v1);
              }
              asm volatile("" : "+v"(v0), "+v"(v1));   // the ROUNDED fp32 values: no contraction into the residuals below
              union { bf16x2_t b; uint32_t u; } H, L;
              H.b = __builtin_convertvector((f32x2_t){v0, v1}, bf16x2_t);
              const float r0 = v0 - __uint_as_float(H.u << 16), r1 = v1 - __uint_as_float(H.u & 0xffff0000u);
              L.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
              const uint32_t m0 = __builtin_amdgcn_perm(L.u, H.u, 0x05040100u), m1 = __builtin_amdgcn_perm(L.u, H.u, 0x07060302u);   // hi | lo << 16
              const uint32_t x0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m0, 0xB1, 0xf, 0xf, false);   // the neighbouring lane's
              const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m1, 0xB1, 0xf, 0xf, false);
              *reinterpret_cast<uint32_t*>(obase + (off0 + 128u * j)) = __builtin_amdgcn_perm(x0, m0, sel_lane);
              *reinterpret_cast<uint32_t*>(obase + (off1 + 128u * j)) = __builtin_amdgcn_perm(x1, m1, sel_lane);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint4 o = o4[r >> 2];
            const uint32_t off = ((r & 3) == 0 ? o.x : (r & 3) == 1 ? o.y : (r & 3) == 2 ? o.z : o.w) + cb;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const float v = finish(acc[i][j][r] + bv[j]);
#if EVT_ABLATE == 6   // timing experiment: no output stores (the arithmetic stays)
              if (v == 12345.678f)
#endif
              *reinterpret_cast<float*>(obase + (off + 128u * j)) = v;
            }
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
        const bool rowok = off != ~0u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          // (hl32 output needs whole 32-column groups: the launcher guarantees Nout % 32 == 0, and the swap runs in all lanes)
          const bool colok = OPL ? (bn * TBN + wn * (NJ * 32) + 32 * j < g.Nout) : (col0 + 32 * j < g.Nout);
          const float v = finish(acc[i][j][r] + bv[j]);
          if (colok && rowok) *reinterpret_cast<float*>(obase + (off + cb + 128u * j)) = v;
        }
      }
    }
  };
  // Raw barrier: __syncthreads() would also drain the global loads and stores that are meant to stay in flight across it.
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // The k-tiles of all my tiles form ONE stream x = 0 .. total - 1 (tile x / nk, k-tile x % nk); LDS stage and register set
  // x & 1.  Iteration i stages x = i + 1, fetches x = i + 3 into the registers just freed, and multiplies x = i (group 0) or
  // x = i + 1 (group 1, which therefore runs "multiply, then stage" between two barriers).  Every wave passes one barrier per
  // iteration: barrier b_i separates all reads of stream element i from the writes of element i + 2 into the same stage, and
  // all writes of i + 1 from its reads.  The stream does not stop at a tile boundary: a finished tile is stored (epilogue) in
  // front of the wave's first multiply of the next tile, whose loads are already in flight -- no prologue / drain bubble per
  // tile.
  // Code-generation notes: ONE fetch / stage / multiply site per register set, and the fetch unconditional (the last
  // iterations re-read the last k-tile) -- with a prologue fetch, a conditional fetch or a multiply per wave group hipcc
  // merged register copies at the joins: twice the accumulators, the prefetched weights parked in scratch, or freshly loaded
  // registers moved at the back edge, each of which waits for the loads right after issuing them.
  issue_indices(0);
  enter_tile(0);
  st_bn = bn_s;
  st_upd = bn_s;
  st_m0 = m0_s;
#pragma unroll
  for (int j = 0; j < AJ; ++j) st_off[j] = a_off[j];
  int sk = 0, sseq = 0;    // load side: k-tile / tile sequence number of the stream element fetched last
  int mk = stage_first ? -(DEPTH + 1) : -DEPTH, mseq = 0;   // multiply side: k-tile / tile of the element this wave multiplies in iteration i
  auto step = [&](Regs& R, int s, int i) __attribute__((always_inline)) {   // s = (i + 1) & 1, a compile-time constant at both call sites
    EVT_TICK(7);
    // The prefetched indices of the next tile are loop-carried registers that a load may have written: pinned HERE, at the
    // top of the iteration (the staging below waits for every older load anyway), they count as clean afterwards.
    // Otherwise hipcc puts s_waitcnt vmcnt(0) in front of the register moves that merge them further down -- behind the
    // gate-reference stores of the staging: the next fetch then waited for those stores to be acknowledged, ~1 us on every
    // k-tile that refreshes p.
    if (DEPTH == 1) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) asm volatile("" : "+v"(nsrc[j]));
      asm volatile("" : "+v"(nrow), "+v"(nbias));
    }
    if (i >= -1 && i + 1 < total) stage(R, s);
    EVT_TICK(0);   // staging (incl. the wait for the loads)
    if (i >= -DEPTH && i + 1 + DEPTH < total) {
      if (sk + 1 == nk) {
        enter_tile(++sseq);
        sk = 0;
      } else {
        ++sk;
        if (sk == nk - 1 && sseq + 1 < ntile) issue_indices(sseq + 1);
      }
    }
    fetch(R, sk);
    EVT_TICK(1);   // tile bookkeeping + issuing the loads
    if (i >= -1 && (PP || !stage_first || i < 0)) barrier();
    EVT_TICK(2);   // barrier in front of the multiply (group 1)
    if (mk >= 0 && mseq < ntile) {
      if (mk == 0 && mseq > 0) {
        epilogue(mseq - 1);
        clear();
        EVT_TICK(3);   // epilogue
      }
      multiply((mseq * nk + mk) & 1);
    }
    EVT_TICK(4);   // multiply
    if (++mk == nk) { mk = 0; ++mseq; }
    if (i >= 0 && (PP || stage_first)) barrier();
    EVT_TICK(5);   // barrier behind the multiply (group 0)
  };
  if (PP && !stage_first) barrier();   // group 1 runs one barrier interval behind group 0
  if (DEPTH == 2) {
    for (int i = -3; i < total; i += 2) {
      step(R0, 0, i);       // stream element i + 1 is even
      step(R1, 1, i + 1);
    }
  } else {
    for (int i = -2; i < total; ++i) step(R0, (i + 1) & 1, i);
  }
  if (PP && stage_first) barrier();
  epilogue(ntile - 1);
#ifdef EVT_PROF
  EVT_TICK(3);
  if (prof_on && lane == 0) {
    for (int q = 0; q < 8; ++q) evt_prof_buf[wave == 0 ? 0 : 1][q] = prof_acc[q];
  }
#endif
}

template <int ACT, int TBM, int TBN, int WM, int WN, int DEPTH, int FMT>
void launch_big_one(const LinArgs& a, hipStream_t s, dim3 grid, int tiles_n, int tiles_total) {
  constexpr size_t lds_bytes = (size_t)2 * (2 * TBM * 32 + 2 * TBN * 32 + 32 + ((FMT & 1) ? 32 : 0)) * 2 + (size_t)2 * TBM * 4 + (size_t)2 * TBN * 4;
  EVT_ALLOW_LDS((gated_linear_split_big_kernel<ACT, TBM, TBN, WM, WN, DEPTH, FMT>), lds_bytes);   // once per instantiation and device
  hipLaunchKernelGGL((gated_linear_split_big_kernel<ACT, TBM, TBN, WM, WN, DEPTH, FMT>), grid, dim3(WM * WN * 64), lds_bytes, s, a, tiles_n,
                     tiles_total);
}

template <int TBM, int TBN, int WM, int WN, int DEPTH>
void launch_big_cfg(const LinArgs& a, hipStream_t s) {
  const int M = a.B * a.kcap;
  const int tiles_m = (M + TBM - 1) / TBM, tiles_n = (a.Nout + TBN - 1) / TBN;
  const dim3 grid(std::min(tiles_m * tiles_n, evt_cu_count()));   // persistent: one workgroup per CU
  const int tt = tiles_m * tiles_n;
  // formats in use: fp32 -> fp32 (any activation), fp32 -> hl32 with GELU (first half of the MLP), hl32 -> fp32 (second half)
  if (a.a_bf16) launch_big_one<EVT_ACT_NONE, TBM, TBN, WM, WN, 1, 4>(a, s, grid, tiles_n, tt);
  else if (a.a_planes) launch_big_one<EVT_ACT_NONE, TBM, TBN, WM, WN, 1, 1>(a, s, grid, tiles_n, tt);   // (one register set: room for both fragment sets)
  else if (a.out_planes) launch_big_one<EVT_ACT_GELU_ERF, TBM, TBN, WM, WN, DEPTH, 2>(a, s, grid, tiles_n, tt);
  else if (a.act == EVT_ACT_GELU_ERF) launch_big_one<EVT_ACT_GELU_ERF, TBM, TBN, WM, WN, DEPTH, 0>(a, s, grid, tiles_n, tt);
  else launch_big_one<EVT_ACT_NONE, TBM, TBN, WM, WN, DEPTH, 0>(a, s, grid, tiles_n, tt);
}

}  // namespace

// Picks a 256-row tile when the launch has enough of them to fill the chip; 0 when the 128x128 kernel (or its split-K form)
// should run instead.  EVT_GEMM_BIG: 0 never, 1 (default) automatic, 2 always 256x256, 3 always 256x128, 4 always 256x192.
int evt_big_choice(const LinArgs& a) {
  static const int mode = getenv("EVT_GEMM_BIG") ? atoi(getenv("EVT_GEMM_BIG")) : 1;
  // whole 32-k tiles, at least two of them; top-k gating only (the threshold policy's masked rows stay with the 128x128 kernel,
  // which skips dead tiles)
  if (mode == 0 || a.Wsplit == nullptr || (a.K & 31) != 0 || a.K < 64 || a.count != nullptr) return 0;
  // 32-bit byte offsets inside the kernel: activations (and the gate reference, same shape), weight planes and output below 4 GB
  // (the activation bound is taken at 4 bytes per element also for a bf16 launch: it then covers the fp32 gate reference)
  if ((int64_t)a.B * a.a_rows * a.lda * 4 >= ((int64_t)1 << 32) ||
      (int64_t)a.Nout * hl32_pitch(a.K) * 2 >= ((int64_t)1 << 32) ||
      (int64_t)a.B * a.o_rows * a.ldo * 4 >= ((int64_t)1 << 32))
    return 0;
  if (mode >= 2 && mode <= 4) return mode;
  // One persistent workgroup per CU: a launch of T tiles runs in ceil(T / CUs) rounds.  Take the widest tile whose columns
  // divide Nout (no wasted edge columns) and whose last round is at least 85 % full; measured at M = 32768 (B = 256 clips):
  // 256x256 for Nout = 2304 / 3072 (343 / 474 us vs 380 / 531 for the 128x128 kernel), 256x192 for Nout = 768 (126 / 418 vs
  // 134 / 443 us; 256x256 would leave a quarter of the CUs idle in its second round).
  const int cus = evt_cu_count(), M = a.B * a.kcap;
  const int tiles_m = (M + 255) / 256;
  auto fills = [&](int tbn) {
    if (a.Nout % tbn != 0) return false;
    const int tiles = tiles_m * (a.Nout / tbn), rounds = (tiles + cus - 1) / cus;
    return tiles >= cus && tiles * 100 >= rounds * cus * 85;
  };
  // Round 5 (evt_linear_pipe.hip, 8 waves): 256x192 first -- QKV (Nout = 2304) is 1536 tiles = 6.0 rounds instead of the 4.5 (-> 5) of
  // 256x256, and the 192-column instantiations keep every register: 310 vs 346 us for QKV, 441 vs 497 for MLP-1 + GELU, 790 vs 859
  // for the MLP pair at B = 256 (profiles/r05/gemm_tile_shapes.txt).
  static const int pipe = getenv("EVT_GEMM_PIPE") ? atoi(getenv("EVT_GEMM_PIPE")) : 1;
  if (pipe && fills(192)) return 4;
  if (fills(256)) return 2;
  if (fills(192)) return 4;
  return 0;
}

bool evt_launch_split_big(const LinArgs& a, hipStream_t s) {
  static const int pipe = getenv("EVT_GEMM_PIPE") ? atoi(getenv("EVT_GEMM_PIPE")) : 1;
  const int choice = evt_big_choice(a);
  if (pipe && choice != 0 && evt_launch_split_pipe(a, s, choice)) return true;
  switch (choice) {
    case 2: launch_big_cfg<256, 256, 4, 2, 1>(a, s); return true;
    case 3: launch_big_cfg<256, 128, 4, 2, 1>(a, s); return true;
    case 4: launch_big_cfg<256, 192, 4, 2, 1>(a, s); return true;   // (a two-deep register prefetch measured the same: kept out)
    default: return false;
  }
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_buf), sizeof(unsigned long long) * 16);
}
#endif
