// evt_linear_small.hip -- K3/K7 for SMALL gated row counts (one video stream: M = k = 256 rows at ViTDet 672^2, a few
// hundred live rows of kcap = N under the threshold policy; ViViT at a handful of clips): a latency-oriented
// split-precision gated linear.
//
// Why a third kernel.  At M = 256 the 128x128 kernel (evt_linear.hip) has 2 x tiles_n tiles, so it splits K over up to 16
// workgroups per tile, each walking its k-tiles with ONE tile of prefetch behind two barriers, and a second launch
// (splitk_finish_kernel) sums the partial planes: 14-18 us + 5.5 us per gated linear, ~50 of them per frame = 1.0 of the
// 2.4 ms a 672^2 frame takes (rocprofv3 trace, round 3).  Nothing in it is bandwidth- or MFMA-bound: the launch is a
// chain of dependent round trips (index list -> row pointers -> first tile -> ... -> partials -> finish).
//
// This kernel:
//   * small output tiles (64x64 or 32x32) so that a launch has 140-200 workgroups WITHOUT splitting K across workgroups
//     (QKV 4 x 36, MLP-1 4 x 48, projection and MLP-2 8 x 24): no partial planes, no finish pass;
//   * the eight waves of a workgroup split the workgroup's K range between them (each wave owns the whole tile over an
//     eighth of K: three k-tiles of 32 for K = 768) and the partial accumulators are summed through LDS at the end -- a split-K whose reduction never
//     leaves the CU;
//   * with K split by wave the waves share no operand bytes: every wave stages ITS k-tiles through a private 16 KB LDS
//     region (coalesced loads of whole 128-byte lines -- A rows gathered through the gate's index list and split into
//     bf16 hi / lo while stored, weight rows as hl32 lines -- then MFMA fragment reads), the next k-tile's loads in
//     flight during the multiply.  LDS operations of one wave execute in order, so there is NO barrier in the k loop;
//   * weight requests are issued before the index list is read: the only dependent chain is index -> A rows.
// Arithmetic is that of gated_linear_split_kernel (A_lo.W_hi + A_hi.W_lo + A_hi.W_hi on v_mfma_f32_32x32x16_bf16, fp32
// accumulate); the k order inside a wave is ascending and the four wave partials are added in wave order, so results
// are deterministic (they differ from the 128x128 kernel's by fp32 summation order only).
#include "evt_linear.h"
#include <stdlib.h>
#include <algorithm>

namespace {

typedef unsigned int u32x4s_t __attribute__((ext_vector_type(4)));

// PARTIAL: the workgroup contracts k-tiles [split * kps, ...) only and writes the raw sums to workspace plane `split`
// (compact rows m); bias / activation / scatter are left to splitk_finish_kernel.
// NW: waves per workgroup = ways the workgroup's K range is split.  LOOPED: the workgroup walks row tiles bm, bm + gm, ...
// below a device-side count (one stream under the threshold policy); the straight-line form keeps the 64x64 tile within
// the 256 registers two waves per SIMD allow (the loop spilled 47 of them: 12.2 -> 18.8 us for QKV at M = 256), so the
// looped form runs four waves.
template <int ACT, int BM, int BN, bool PARTIAL, int NW, bool LOOPED>
__global__ __launch_bounds__(64 * NW, LOOPED ? 2 : 1) void gated_linear_small_kernel(const LinArgs g, int tiles_n, int ksplit, int gm) {
  constexpr int MI = BM / 32, NJ = BN / 32, SMALL_THREADS = 64 * NW;
  constexpr int AJ = BM / 8, WJ = BN / 8;                 // 16-byte pieces per lane per k-tile: 8 rows x 8 chunks per wave instruction
  constexpr int WSTAGE = 2 * BM * 32 + 2 * BN * 32;       // bf16 elements of one wave's stage: A hi, A lo, W hi, W lo
  static_assert((size_t)WSTAGE * 2 >= (size_t)BM * BN * 4, "a wave's partial tile reuses its operand stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char evt_small_smem[];
  int64_t* orow_off = reinterpret_cast<int64_t*>(evt_small_smem + (size_t)NW * WSTAGE * 2);   // [BM] output row offset (elements), -1 = masked row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __bf16* Ahi = reinterpret_cast<__bf16*>(evt_small_smem) + wave * WSTAGE;   // this wave's private operand stage
  __bf16* Alo = Ahi + BM * 32;
  __bf16* Bhi = Alo + BM * 32;
  __bf16* Blo = Bhi + BN * 32;
  float* red = reinterpret_cast<float*>(evt_small_smem);   // [NW][BM][BN] fp32: wave w's partial tile at its own stage
  // 64-byte LDS rows, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3) (see gated_linear_split_kernel)
  auto lds_off = [](int row, int k) { return row * 32 + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); };
  const int lr = lane & 31, lh = lane >> 5;
  const int split = PARTIAL ? (int)blockIdx.z : 0;
  const int M = g.B * g.kcap;
  // XCD-aware tile order (workgroup w runs on XCD w % 8): all row tiles of a column tile run on ONE XCD, so a weight tile
  // is fetched into one private L2 (two for a tile that straddles a run boundary) instead of one per row tile; the grid is
  // padded to a multiple of 8 and the surplus workgroups leave.  gm = row tiles launched: all of them, or -- one
  // stream under the threshold policy, kcap = N rows of which count[0] are live -- enough for 512 rows, each workgroup
  // walking row tiles bm, bm + gm, ... below the device-side count (a grid over all kcap / BM row tiles would start
  // thousands of 512-thread workgroups only to retire them).
  // (The (column tile, row tile) pairs in column-major order are dealt to the XCDs in eight CONTIGUOUS, equally long runs:
  // "column tile bn on XCD bn % 8" gave XCDs 0-3 five column tiles and XCDs 4-7 four at Nout = 2304 -- 35 workgroups on a
  // 32-CU XCD with 7 row tiles, i.e. a second round for three of them: QKV at M = 409 took 22.5 us against 12.7 at M = 256.)
  // Under a device-side count (LOOPED) the launch does not know how many of its gm row tiles are live: row-major order
  // instead, column tile bn on XCD bn % 8 (the column count padded to a multiple of 8) -- the live row tiles are dispatched
  // first and every XCD gets the same share of each; the dead ones retire behind them (with 13 of 16 row tiles live, the
  // column-major runs cost QKV 18.5 us against 15.9 us for a launch of exactly 13).
  int bn, bm_first;
  if (LOOPED) {
    const int tn8 = (tiles_n + 7) & ~7;
    bm_first = (int)blockIdx.x / tn8;
    bn = (int)blockIdx.x - bm_first * tn8;
    if (bn >= tiles_n) return;
  } else {
    const int x8 = blockIdx.x & 7, s8 = blockIdx.x >> 3;
    const int total = tiles_n * gm, q8 = total >> 3, r8 = total & 7;
    if (s8 >= q8 + (x8 < r8 ? 1 : 0)) return;
    const int t8 = x8 * q8 + min(x8, r8) + s8;
    bn = t8 / gm;
    bm_first = t8 - bn * gm;
  }
  const int rows_live = LOOPED ? min(M, g.count[0]) : M;
  const int n0_inv = bn * BN;
  for (int bm = bm_first; bm * BM < rows_live; bm += gm) {
  const int m0 = bm * BM;
  // (LOOPED: the column offset and the lane index are made opaque per iteration -- hoisted out of the row-tile loop, everything
  // derived from them stays live across the epilogue, and the 64x64 variant needed 256 VGPR + 116 AGPR: one wave per SIMD)
  int n0 = n0_inv, lane_v = lane;
  if (LOOPED) { asm volatile("" : "+s"(n0)); asm volatile("" : "+v"(lane_v)); }

  // k range of this workgroup (whole 32-k tiles), then of this wave
  const int nk_all = g.K / 32, kps = (nk_all + ksplit - 1) / ksplit;
  const int t_lo = split * kps, t_hi = min(nk_all, t_lo + kps);
  const int per_wave = (t_hi - t_lo + NW - 1) / NW;
  const int t0 = t_lo + wave * per_wave, t1 = min(t_hi, t0 + per_wave);
  const int tc0 = t0 < t1 ? t0 : (t_lo < nk_all ? t_lo : 0);   // clamped: every load unconditional

  // ---- staging assignment: lane -> row (lane >> 3) + 8 j, 16-byte chunk lane & 7 of the row's 128-byte k-tile.  One wave
  // instruction reads 8 whole 128-byte lines.  (Row-per-lane loads straight into MFMA fragments -- 32 lines per
  // instruction, each touched by four instructions -- made the launch L1-tag-bound: 21 us for QKV at M = 256.)
  const int sr = lane_v >> 3, sc = lane_v & 7;
  const int64_t wpitch = hl32_pitch(g.K);
  // 32-bit byte offsets from the (scalar) base pointers: the 64x64 tile sits at the register limit of two waves per SIMD
  // (the launcher guarantees activations and weight planes below 2 GB)
  const char* const Wbase = reinterpret_cast<const char*>(g.Wsplit);
  const char* const Abase = reinterpret_cast<const char*>(g.A);
  uint32_t wsrc[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int n = n0 + sr + 8 * j;
    wsrc[j] = (uint32_t)((n < g.Nout ? n : g.Nout - 1) * (int)wpitch + sc * 8) * 2u;   // chunks 0-3: hi plane, 4-7: lo plane
  }
  struct Regs { f32x4 a[AJ]; u32x4s_t w[WJ]; };   // one k-tile of this wave's operands in flight
  auto fetch_w = [&](Regs& R, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < WJ; ++j) R.w[j] = *reinterpret_cast<const u32x4s_t*>(Wbase + (wsrc[j] + (uint32_t)t * 128u));
  };
  Regs R0, R1;
  const int tc1 = t0 + 1 < t1 ? t0 + 1 : tc0;
  fetch_w(R0, tc0);   // weights first: they depend on nothing
  fetch_w(R1, tc1);

  // ---- rows of the tile: liveness, output offsets, A row pointers (the dependent chain: index list -> rows) ---------
  int live = 0;
  uint32_t asrc[AJ];
  bool aok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) { aok[j] = false; asrc[j] = (uint32_t)sc * 16u; }
  for (int r = tid; r < BM; r += SMALL_THREADS) {
    const int m = m0 + r;
    int64_t off = -1;
    if (m < M) {
      const int b = m / g.kcap, ii = m - b * g.kcap;
      if (g.count == nullptr || ii < g.count[b]) {
        const int o = (g.o_idx != nullptr) ? g.o_idx[m] : ii;
        off = ((int64_t)b * g.o_rows + o) * g.ldo;
        live |= 1;
      }
    }
    orow_off[r] = off;
  }
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + sr + 8 * j;
    if (m >= M) continue;
    const int b = m / g.kcap, ii = m - b * g.kcap;
    if (!(g.count == nullptr || ii < g.count[b])) continue;
    const int src = (g.a_idx != nullptr) ? g.a_idx[m] : ii;
    asrc[j] = (uint32_t)((b * g.a_rows + src) * (int)g.lda + sc * 4) * 4u;
    aok[j] = true;
  }
  // threshold policy: kcap = N but only count[b] rows are live -- tiles of masked rows have nothing to do (with PARTIAL
  // their workspace rows stay unwritten; splitk_finish_kernel skips the same rows)
  if (!__syncthreads_or(live)) return;
  auto fetch_a = [&](Regs& R, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) R.a[j] = *reinterpret_cast<const f32x4*>(Abase + (asrc[j] + (uint32_t)t * 128u));
  };
  fetch_a(R0, tc0);
  fetch_a(R1, tc1);

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // gate reference refresh (K2 fused): every column tile stages the same A rows; column tile bn writes back the k-tiles
  // t with t % tiles_n == bn (the wave that owns t does), from the fp32 registers it holds anyway
  const bool do_upd = g.p_upd != nullptr;
  // The stage is PRIVATE to the wave: LDS operations of one wave execute in order, so the fragment reads of k-tile t are
  // complete (their MFMAs have consumed them) before the stores of k-tile t + 1 are issued -- no barrier, one stage.
  // Two register sets used alternately (TWO k-tiles in flight: a wave owns ~3, so it waits for about two round trips in
  // all), every fetch unconditional (clamped k-tile) so that the waits can count on the younger requests being in flight.
  auto step = [&](Regs& R, int t) __attribute__((always_inline)) {
    // registers -> stage (activations split into bf16 hi / lo on the way)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      bf16x4_t h, l;
      split4(make_float4(R.a[j].x, R.a[j].y, R.a[j].z, R.a[j].w), &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + lds_off(sr + 8 * j, sc * 4)) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + lds_off(sr + 8 * j, sc * 4)) = l;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j)
      *reinterpret_cast<u32x4s_t*>(((sc & 4) ? Blo : Bhi) + lds_off(sr + 8 * j, (sc & 3) * 8)) = R.w[j];
    if (do_upd && (t % tiles_n) == bn) {
#pragma unroll
      for (int j = 0; j < AJ; ++j)
        if (aok[j]) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(g.p_upd) + (asrc[j] + (uint32_t)t * 128u)) = R.a[j];
    }
    { const int tn = t + 2 < t1 ? t + 2 : tc0; fetch_w(R, tn); fetch_a(R, tn); }   // the k-tile after next, into the set just drained
    // other LANES of this wave wrote the fragments read below: the stores must have been performed (in-order LDS, one wait)
    // and the compiler must not move LDS accesses across this point
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 32; ks += 16) {
      bf16x8_t ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ao = lds_off(32 * i + lr, ks + 8 * lh);
        ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao);
        al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int bo = lds_off(32 * j + lr, ks + 8 * lh);
        bh[j] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo);
        bl[j] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("" ::: "memory");   // (the next k-tile's stage stores stay behind these fragment reads)
    __builtin_amdgcn_wave_barrier();
  };
  for (int t = t0; t < t1; t += 2) {
    step(R0, t);
    if (t + 1 < t1) step(R1, t + 1);
  }

  // ---- the waves' partial tiles -> LDS (each over its own, now idle, stage); wave-order sum, bias, activation, scatter ----
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* mine = reinterpret_cast<float*>(Ahi);
  constexpr int RS = WSTAGE / 2;   // floats between two waves' partial tiles
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        mine[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + 32 * j + lr] = acc[i][j][r];
  __syncthreads();
  constexpr int C4 = BN / 4;                                        // float4 columns per row
  constexpr int PER = (BM * C4 + SMALL_THREADS - 1) / SMALL_THREADS;   // float4 pieces per thread
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int e = tid + SMALL_THREADS * it, row = e / C4, c = (e - row * C4) * 4;
    if (e >= BM * C4) break;
    float4 v = *reinterpret_cast<const float4*>(red + row * BN + c);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(red + w * RS + row * BN + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int n = n0 + c, m = m0 + row;
    if (PARTIAL) {
      if (m < M && n < g.Nout) *reinterpret_cast<float4*>(g.ws + ((int64_t)split * M + m) * g.Nout + n) = v;   // Nout % 4 == 0 (launcher)
      continue;
    }
    const int64_t off = orow_off[row];
    if (off < 0 || n >= g.Nout) continue;
    const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    if (ACT == EVT_ACT_GELU_ERF) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
    *reinterpret_cast<float4*>(g.out + off + n) = v;
  }
  if (!LOOPED) return;
  __syncthreads();   // the next row tile reuses the stages and the row table
  }
}

template <int ACT, int BM, int BN, int NW, bool LOOPED>
void launch_small_inst(const LinArgs& a, hipStream_t s, int ksplit, int tiles_n, int gm) {
  const size_t stages = (size_t)NW * (2 * BM * 32 + 2 * BN * 32) * 2;
  const size_t lds = stages + (size_t)BM * sizeof(int64_t);
  const dim3 grid(LOOPED ? ((tiles_n + 7) & ~7) * gm : 8 * ((tiles_n * gm + 7) / 8), 1, ksplit);
  if (ksplit > 1) {
    EVT_ALLOW_LDS((gated_linear_small_kernel<ACT, BM, BN, true, NW, LOOPED>), lds);
    hipLaunchKernelGGL((gated_linear_small_kernel<ACT, BM, BN, true, NW, LOOPED>), grid, dim3(64 * NW), lds, s, a, tiles_n, ksplit, gm);
  } else {
    EVT_ALLOW_LDS((gated_linear_small_kernel<ACT, BM, BN, false, NW, LOOPED>), lds);
    hipLaunchKernelGGL((gated_linear_small_kernel<ACT, BM, BN, false, NW, LOOPED>), grid, dim3(64 * NW), lds, s, a, tiles_n, 1, gm);
  }
}

template <int ACT, int BM, int BN>
void launch_small_tile(const LinArgs& a, hipStream_t s, int ksplit) {
  const int M = a.B * a.kcap;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (a.Nout + BN - 1) / BN;
  const bool looped = a.count != nullptr && a.B == 1;
  // row tiles launched under a device-side count: enough for 512 rows, but not more (column, row) pairs than CUs when that
  // still covers 256 rows -- a 9th workgroup on a 32-CU XCD is a second round
  const int gm_l = std::min(tiles_m, 512 / BM);
  const int gm = looped ? gm_l : tiles_m;
  if (looped) launch_small_inst<ACT, BM, BN, 4, true>(a, s, ksplit, tiles_n, gm);
  else launch_small_inst<ACT, BM, BN, 8, false>(a, s, ksplit, tiles_n, gm);
}

}  // namespace

// Takes the launch when it is small (see the header); returns the K split it used (0: not taken).  With a split > 1 the
// caller runs splitk_finish_kernel over the workspace planes.  EVT_GEMM_SMALL=0 turns the path off.
bool evt_small_accepts(const LinArgs& a) {
  static const int on = getenv("EVT_GEMM_SMALL") ? atoi(getenv("EVT_GEMM_SMALL")) : 1;
  if (!on || a.Wsplit == nullptr || (a.K & 31) != 0 || (a.Nout & 3) != 0 || (a.ldo & 3) != 0 || (a.lda & 3) != 0) return false;
  const int64_t M = (int64_t)a.B * a.kcap;
  // live rows: all of M for top-k; with a per-clip count (threshold policy) kcap = N but few rows are live -- the dead tiles
  // exit at once.  Larger launches belong to the 128x128 / 256-row kernels.
  if ((int64_t)a.B * a.a_rows * a.lda * 4 >= ((int64_t)1 << 31) || (int64_t)a.Nout * hl32_pitch(a.K) * 2 >= ((int64_t)1 << 31)) return false;
  const bool counted = a.count != nullptr;
  if (!counted && M > 1024) return false;
  if (counted && (a.B > 4 || M > 16384)) return false;
  return evt_big_choice(a) == 0;
}

int evt_launch_split_small(const LinArgs& a, hipStream_t s) {
  if (!evt_small_accepts(a)) return 0;
  const int64_t M = (int64_t)a.B * a.kcap;
  const bool counted_ = a.count != nullptr;
  const int live = counted_ ? (int)std::min<int64_t>(M, 512) : (int)M;   // planning figure for the tile choice
  // tile: 64x64 unless that leaves fewer than ~128 workgroups (graph-replayed launches at M = 256, us incl. the ~1.5 us
  // boundary, 64x64 / 32x32: QKV 12.2 / 14.0, MLP-1 12.7 / 16.0, projection 11.3 / 6.6, MLP-2 24.8 / 13.1; MLP-2 as
  // 64x64 with K split four ways over workgroups + finish pass: 16.5).  No K split over workgroups.
  int bm = 64, ks = 1;
  auto wgs = [&](int b) { return ((live + b - 1) / b) * ((a.Nout + b - 1) / b); };
  if (wgs(64) < 128) bm = 32;
  // One stream under a device-side count: the row-tile loop costs the 64x64 variant its occupancy (256 VGPR + 116 AGPR with four
  // waves = 4 waves per CU; with eight waves it spills) -- at count = 409 of 4096 rows QKV 28.4 us / MLP-1 28.3 us against
  // 20.4 / 22.2 us on 32x32 tiles (146 + 16 registers, three workgroups per CU).
  if (counted_ && a.B == 1) bm = 32;
  const bool gelu = a.act == EVT_ACT_GELU_ERF;
  if (bm == 64) {
    if (gelu) launch_small_tile<EVT_ACT_GELU_ERF, 64, 64>(a, s, ks);
    else launch_small_tile<EVT_ACT_NONE, 64, 64>(a, s, ks);
  } else {
    if (gelu) launch_small_tile<EVT_ACT_GELU_ERF, 32, 32>(a, s, ks);
    else launch_small_tile<EVT_ACT_NONE, 32, 32>(a, s, ks);
  }
  return ks;
}
