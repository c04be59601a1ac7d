// evt_linear.hip -- K3/K7: gated linear and gated MLP on the matrix cores.
//
//   out[orow(m), n] = act( sum_k A[arow(m), k] * W[n, k] + bias[n] )
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate; bitwise an fmaf chain), so fp32
// activations agree with the reference's fp32 `addmm` to rounding-order noise (~1e-6).
//
// Tiling (64-wide wavefronts): 128x128 output tile per 256-thread workgroup, BK = 32; the four
// waves form a 2x2 grid of 64x64 wave tiles, each a 2x2 grid of 32x32 MFMA accumulators
// (64 accumulator registers per lane).  Both operands are K-contiguous in HBM (A rows are token
// rows, W rows are output features), so both tiles are staged with 16-byte loads of whole 128-byte
// row segments; the A row pointers go through the gate's index list, i.e. the gather of active
// tokens happens while the tile is written into LDS and no compact copy of the rows ever exists.
// LDS rows are padded to 36 floats: the 16 rows a ds_read_b128 lane group touches then start 36
// banks apart and tile all 64 banks (conflict-free).  The MFMA k-index is permuted -- lane half
// h reads k in [16h, 16h+16) as four b128 reads instead of sixteen strided b32 reads; a sum over k
// does not care which k goes with which MFMA step as long as A and B agree.
// Global->register prefetch of tile t+1 is issued before the MFMAs of tile t (one barrier per
// k-tile, two LDS buffers).  The epilogue adds bias, applies GELU if asked (erff in this exact-fp32 kernel; the rational
// erf of evt_linear.h, |error| < 2e-6, in the split-precision kernels), and scatters
// rows through o_idx (TokenBuffer update fused); column-block 0 also refreshes the gate reference
// rows (p_upd) from the A tile it already holds.
#include "evt_linear.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;  // LDT: padded LDS row (floats)
constexpr int GEMM_THREADS = 256;

#ifndef EVT_GEMM_PRIO
#define EVT_GEMM_PRIO 1
#endif
#ifndef EVT_GEMM_MIN_BLOCKS
#define EVT_GEMM_MIN_BLOCKS 3   // 158 VGPRs, 3 x 42 KB LDS: three workgroups per CU (+8 % over two; measured)
#endif
constexpr int EVT_SPLITK_DYN_MAX_CLIPS = 32;  // device-side split-K choice reads every clip's count per workgroup

template <int ACT>
__global__ __launch_bounds__(GEMM_THREADS) void gated_linear_kernel(const LinArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDT];
  auto As = [&](int buf) { return lds + buf * (BM + BN) * LDT; };
  auto Bs = [&](int buf) { return lds + buf * (BM + BN) * LDT + BM * LDT; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int M = g.B * g.kcap;
  __shared__ int64_t orow_off[BM];  // output row offset (elements) of each tile row, -1 = masked row
  for (int r = tid; r < BM; r += GEMM_THREADS) {
    const int m = m0 + r;
    int64_t off = -1;
    if (m < M) {
      const int b = m / g.kcap, ii = m - b * g.kcap;
      if (g.count == nullptr || ii < g.count[b])
        off = ((int64_t)b * g.o_rows + ((g.o_idx != nullptr) ? g.o_idx[m] : ii)) * g.ldo;
    }
    orow_off[r] = off;
  }

  // ---- staging assignment: thread -> (row r0 + 32*j, 16-byte column c4) for j = 0..3
  const int r0 = tid >> 3, c4 = tid & 7;
  const float* a_ptr[4];
  const float* w_ptr[4];
  bool a_ok[4], w_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + r0 + 32 * j;
    a_ok[j] = false;
    a_ptr[j] = g.A;
    if (m < M) {
      const int b = m / g.kcap, i = m - b * g.kcap;
      if (g.count == nullptr || i < g.count[b]) {
        const int src = (g.a_idx != nullptr) ? g.a_idx[m] : i;
        a_ptr[j] = g.A + ((int64_t)b * g.a_rows + src) * g.lda;
        a_ok[j] = true;
      }
    }
    const int n = n0 + r0 + 32 * j;
    w_ok[j] = n < g.Nout;
    w_ptr[j] = g.W + (int64_t)(w_ok[j] ? n : 0) * g.K;
  }
  // threshold policy: kcap = N but only count[b] rows are live -- tiles made of masked rows only have nothing to do
  if (!__syncthreads_or(a_ok[0] | a_ok[1] | a_ok[2] | a_ok[3])) return;

  float4 ra[4], rw[4];
  auto fetch = [&](int k0) {
    const int kc = k0 + c4 * 4;
    const bool kin = kc < g.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = (a_ok[j] && kin) ? *reinterpret_cast<const float4*>(a_ptr[j] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[j] = (w_ok[j] && kin) ? *reinterpret_cast<const float4*>(w_ptr[j] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<float4*>(As(buf) + (r0 + 32 * j) * LDT + c4 * 4) = ra[j];
      *reinterpret_cast<float4*>(Bs(buf) + (r0 + 32 * j) * LDT + c4 * 4) = rw[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (g.K + BK - 1) / BK;
  fetch(0);
  // Gate reference refresh (K2 fused): every column block stages the same A rows; block x writes back the k-tiles
  // t with t % gridDim.x == x (dealt out so that no block waits for store acknowledgements in every k-tile).
  const bool do_upd = g.p_upd != nullptr;
  const int upd_n = gridDim.x, upd_x = blockIdx.x;
  float* u_ptr[4];
  if (do_upd) {
#pragma unroll
    for (int j = 0; j < 4; ++j) u_ptr[j] = g.p_upd + (a_ptr[j] - g.A);
  }
  stage(0);
  if (do_upd && upd_x == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (a_ok[j] && c4 * 4 < g.K) *reinterpret_cast<float4*>(u_ptr[j] + c4 * 4) = ra[j];
  }
  __syncthreads();

  const int lr = lane & 31, lh = lane >> 5;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) fetch((t + 1) * BK);
    // fragments: 16 k-values per lane-half for 2 row-tiles of A and 2 row-tiles of B
    float4 fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* pa = As(cur) + (wm * 64 + i * 32 + lr) * LDT + lh * 16;
      const float* pb = Bs(cur) + (wn * 64 + i * 32 + lr) * LDT + lh * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[i][q] = *reinterpret_cast<const float4*>(pa + q * 4);
        fb[i][q] = *reinterpret_cast<const float4*>(pb + q * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float av = (e == 0) ? fa[i][q].x : (e == 1) ? fa[i][q].y : (e == 2) ? fa[i][q].z : fa[i][q].w;
            const float bv = (e == 0) ? fb[j][q].x : (e == 1) ? fb[j][q].y : (e == 2) ? fb[j][q].z : fb[j][q].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
      }
    }
    if (t + 1 < nk) {
      stage(cur ^ 1);
      if (do_upd && ((t + 1) % upd_n) == upd_x) {
        const int kc = (t + 1) * BK + c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (a_ok[j] && kc < g.K) *reinterpret_cast<float4*>(u_ptr[j] + kc) = ra[j];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // The bias loads are pinned before the store loop (empty asm reading the registers): with a load still pending inside
  // the predicated store blocks the compiler re-waits `vmcnt(0)` in every block, and on gfx9 vmcnt also counts stores
  // -- every store would wait for the previous one to be acknowledged.
  float bv[2];
  int ncol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    ncol[j] = n0 + wn * 64 + j * 32 + lr;
    bv[j] = ncol[j] < g.Nout ? g.bias[ncol[j]] : 0.f;
  }
  asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));   // pin: see above
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t off = orow_off[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      if (off < 0) continue;
      float* orow = g.out + off;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (ncol[j] < g.Nout) {
          float v = acc[i][j][r] + bv[j];   // (activation per element here: doing all 64 up front costs 268 VGPRs -> 1 workgroup/CU)
          if (ACT == EVT_ACT_GELU_ERF) v = gelu_erf_exact(v);
          orow[ncol[j]] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-precision variant: every fp32 operand x is written x = hi + lo + O(2^-17 |x|) with hi, lo
// bf16 (round-to-nearest-even), and a.w is accumulated in fp32 as hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate, 3 instructions per product => ~5x the
// throughput).  The dropped lo.lo term is <= 2^-16 |a||w| per product, random in sign; measured
// against fp64 it is ~1e-5 relative on these shapes (tests/test_gpu_kernels.py), two orders of
// magnitude inside the 1e-3 activation tolerance.  Weights are split once (evt_split_weights);
// activations are split while the gathered A tile is staged into LDS (v_cvt_pk_bf16_f32).
// Same tiling as the fp32 kernel: 128x128x32, 2x2 waves of 2x2 32x32 accumulators, A rows gathered
// through a_idx during staging, scatter epilogue, p_upd refresh by column-block 0.
// ---------------------------------------------------------------------------------------------
// Split-K factor for `tiles` live 128x128 output tiles and nk k-tiles: only when the tile count leaves most of
// the 256 CUs idle; every split keeps >= 4 k-tiles.  Evaluated on the host from the shape, or -- with a
// per-clip count (threshold policy: kcap = N, few live rows) -- on the device from the counts, identically by
// the GEMM workgroups and by splitk_finish_kernel.
__host__ __device__ inline int splitk_for(int tiles, int nk) {
  if (tiles <= 0 || tiles >= 128) return 1;
  int s = (256 + tiles - 1) / tiles;
  s = s < nk / 4 ? s : nk / 4;
  s = s < 16 ? s : 16;
  if (s < 2) return 1;
  const int kps = (nk + s - 1) / s;
  return (nk + kps - 1) / kps;  // drop empty trailing splits
}

// dyn != 0: live tiles = sum_b ceil(count[b] / 128) * tiles_n (exact when kcap % 128 == 0; any consistent
// estimate is correct, it only has to be the same everywhere)
__device__ __forceinline__ int splitk_dynamic(const int32_t* count, int B, int tiles_n, int nk) {
  int live = 0;
  for (int b = 0; b < B; ++b) live += (count[b] + 127) >> 7;
  return splitk_for(live * tiles_n, nk);
}

// Tile configuration: TBM x TBN output tile, TBK k-tile, WM x WN waves each owning a 64x64 sub-tile
// (2x2 MFMA accumulators).  Workgroups are numbered so that one XCD (private L2) walks consecutive
// column tiles of the same row tile: the gathered A rows are fetched into that L2 once.
template <int ACT, int TBM, int TBN, int TBK, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && TBK == 32 ? EVT_GEMM_MIN_BLOCKS : 1)) void gated_linear_split_kernel(const LinArgs g, int tiles_n, int tiles_total, int tile_map, int ksplit, int dyn) {
  constexpr int NT = WM * WN * 64;
  // bf16 LDS tile layout.  TBK == 32: unpadded 64-byte rows with the 16-byte chunk c of row r stored at chunk
  // c ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 lane group touches cover all 16 slots of a 256-byte bank row, and the
  // two rows a 16-lane ds_write_b64 group / 8-lane ds_write_b128 group writes occupy disjoint halves of the 32 write
  // banks.  (The padded 80-byte pitch used before was conflict-free for the reads only: rocprof counted a third of the
  // LDS cycles of this kernel as bank conflicts, all from the staging stores.)  Other TBK: padded pitch, no swizzle.
  constexpr bool SWZ = (TBK == 32);
  constexpr int TSP = SWZ ? TBK : TBK + 8;
  auto lds_off = [](int row, int k) {   // element offset of (row, k); k a multiple of 4
    return SWZ ? row * TSP + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7) : row * TSP + k;
  };
  static_assert(TBM == WM * 64 && TBN == WN * 64, "each wave owns 64x64");
  constexpr int BBUF = 2 * TBN * TSP;   // one weight buffer: hi plane, lo plane
  // ONE dynamic LDS object for everything (operand tiles, then the row-offset table): with a second __shared__ object
  // in the kernel hipcc waits vmcnt(0) -- i.e. for the LDS-DMA just issued -- before the fragment reads of every k-tile.
  extern __shared__ __attribute__((aligned(16))) unsigned char evt_gemm_smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(evt_gemm_smem);
  int64_t* orow_off = reinterpret_cast<int64_t*>(lds + 2 * TBM * TSP + BBUF);  // output row offset (elements) of each tile row, -1 = masked row
  __bf16* Ahi = lds;
  __bf16* Alo = lds + TBM * TSP;
  __bf16* Bhi = lds + 2 * TBM * TSP;
  __bf16* Blo = lds + 2 * TBM * TSP + TBN * TSP;

  const int split = blockIdx.x / tiles_total;
  if (dyn) {  // launched with the largest factor the shape allows; the counts decide how many splits work
    ksplit = splitk_dynamic(g.count, g.B, tiles_n, (g.K + TBK - 1) / TBK);
    if (split >= ksplit) return;
  }
  // XCD-aware tile order (dispatch puts workgroup w on XCD w % 8).  map 0: each XCD owns a contiguous run of
  // row-major tiles (bijective for any tile count).  map 1 (tiles_n even, tiles_m % 4 == 0): each XCD owns a
  // (tiles_m/4) x (tiles_n/2) rectangle, so its half of W stays resident in its 4 MB L2 and every A panel is
  // fetched by 2 XCDs instead of being streamed past all of W.
  int tile;
  {
    const int w = blockIdx.x % tiles_total, x = w % 8, sidx = w / 8;
    if (tile_map == 1) {
      const int mg = tiles_total / tiles_n / 4, ngw = tiles_n / 2;
      const int m = (x >> 1) * mg + sidx / ngw, n = (x & 1) * ngw + sidx % ngw;
      tile = m * tiles_n + n;
    } else {
      const int q = tiles_total / 8, r = tiles_total % 8;
      tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sidx;
    }
  }
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = bm * TBM, n0 = bn * TBN;
  const int M = g.B * g.kcap;

  int live = 0;
  for (int r = tid; r < TBM; r += NT) {
    const int m = m0 + r;
    int64_t off = -1;
    if (m < M) {
      const int b = m / g.kcap, ii = m - b * g.kcap;
      if (g.count == nullptr || ii < g.count[b])
        off = ((int64_t)b * g.o_rows + ((g.o_idx != nullptr) ? g.o_idx[m] : ii)) * g.ldo;
    }
    orow_off[r] = off;
    live |= off >= 0;
  }
  // threshold policy: kcap = N but only count[b] rows are live -- tiles made of masked rows only have nothing to do
  // (with split-K their workspace rows stay unwritten; splitk_finish_kernel skips the same rows)
  if (!__syncthreads_or(live)) return;
  // A staging: float4 chunks, TBK/4 per row; thread -> rows ar0 + j*AROWS
  constexpr int ACH = TBK / 4;           // float4 chunks per A row
  constexpr int AROWS = NT / ACH;        // rows covered per pass
  constexpr int AJ = TBM / AROWS;        // passes
  const int ar0 = tid / ACH, ac4 = tid % ACH;
  const float* a_ptr[AJ];
  bool a_ok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + ar0 + AROWS * j;
    a_ok[j] = false;
    a_ptr[j] = g.A;
    if (m < M) {
      const int b = m / g.kcap, i = m - b * g.kcap;
      if (g.count == nullptr || i < g.count[b]) {
        const int src = (g.a_idx != nullptr) ? g.a_idx[m] : i;
        a_ptr[j] = g.A + ((int64_t)b * g.a_rows + src) * g.lda;
        a_ok[j] = true;
      }
    }
  }
  // W staging: 16-byte chunks (8 bf16) of the hi and lo planes, TBK/8 per row
  constexpr int WCH = TBK / 8;
  constexpr int WROWS = NT / WCH;
  constexpr int WJ = TBN / WROWS;
  // weights in the hl32 layout (evt_linear.h): hi and lo of 32 consecutive k share one 128-byte line; rows are
  // zero-filled up to a multiple of 32, so whole k-tiles can always be read
  const uint16_t* whi = g.Wsplit;
  const uint16_t* wlo = g.Wsplit + 32;
  const int64_t wpitch = hl32_pitch(g.K);
  const int wr0 = tid / WCH, wc8 = (tid % WCH) * 8;

  int64_t w_off[WJ];   // element offset of this thread's weight rows (clamped to the last row past Nout)
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int n = n0 + wr0 + WROWS * j;
    w_off[j] = (int64_t)(n < g.Nout ? n : g.Nout - 1) * wpitch;
  }
  float4 ra[AJ];
  uint4 rwh[WJ], rwl[WJ];
  auto fetch = [&](int k0) {
    const int kc = k0 + ac4 * 4;
    const int kw = hl32_hi(k0 + wc8);
    if (k0 + TBK <= g.K) {
      // Whole k-tile inside K (always, when K % TBK == 0): unconditional loads, no per-load branch.  Masked rows
      // (a_ptr = A) and weight rows past Nout (clamped) contribute to accumulators that the epilogue never stores.
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        rwh[j] = *reinterpret_cast<const uint4*>(whi + w_off[j] + kw);
        rwl[j] = *reinterpret_cast<const uint4*>(wlo + w_off[j] + kw);
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) ra[j] = *reinterpret_cast<const float4*>(a_ptr[j] + kc);
      return;
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      ra[j] = (a_ok[j] && kc < g.K) ? *reinterpret_cast<const float4*>(a_ptr[j] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool kin = k0 + wc8 < g.K;  // K % 8 == 0 is required by the launcher; the planes are zero past K anyway
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      rwh[j] = kin ? *reinterpret_cast<const uint4*>(whi + w_off[j] + kw) : make_uint4(0, 0, 0, 0);
      rwl[j] = kin ? *reinterpret_cast<const uint4*>(wlo + w_off[j] + kw) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      bf16x4_t h, l;
      split4(ra[j], &h, &l);
      *reinterpret_cast<bf16x4_t*>(Ahi + lds_off(ar0 + AROWS * j, ac4 * 4)) = h;
      *reinterpret_cast<bf16x4_t*>(Alo + lds_off(ar0 + AROWS * j, ac4 * 4)) = l;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      *reinterpret_cast<uint4*>(Bhi + lds_off(wr0 + WROWS * j, wc8)) = rwh[j];
      *reinterpret_cast<uint4*>(Blo + lds_off(wr0 + WROWS * j, wc8)) = rwl[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Gate reference refresh (K2 fused).  Every column tile of a row tile stages the same activation rows, so the
  // refresh is dealt out over them: column tile bn writes back the k-tiles t with t % tiles_n == bn.  (All of it
  // in column tile 0 made those tiles -- one in tiles_n -- wait for 4 store acknowledgements in every k-tile:
  // +94 us on the 451 us QKV launch at B = 256.)
  const bool do_upd = g.p_upd != nullptr;
  float* u_ptr[AJ];
  if (do_upd) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) u_ptr[j] = g.p_upd + (a_ptr[j] - g.A);
  }
  // split-K: this workgroup contracts k-tiles [t0, nk) of the tile; ksplit == 1 is the whole K.
  const int nk_all = (g.K + TBK - 1) / TBK, kps = (nk_all + ksplit - 1) / ksplit;
  const int t0 = split * kps, nk = min(nk_all, t0 + kps);
  const int lr = lane & 31, lh = lane >> 5;
  fetch(t0 * TBK);
  for (int t = t0; t < nk; ++t) {
    stage();
    const bool upd_now = do_upd && (t % tiles_n) == bn;
    if (upd_now) {
      const int kc = t * TBK + ac4 * 4;
#pragma unroll
      for (int j = 0; j < AJ; ++j)
        if (a_ok[j] && kc < g.K) *reinterpret_cast<float4*>(u_ptr[j] + kc) = ra[j];
    }
    if (t + 1 < nk) fetch((t + 1) * TBK);   // before the barrier: the requests do not wait for the slowest wave's staging
    __syncthreads();
    if (EVT_GEMM_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < TBK; ks += 16) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ao = lds_off(wm * 64 + i * 32 + lr, ks + 8 * lh);
        const int bo = lds_off(wn * 64 + i * 32 + lr, ks + 8 * lh);
        ah[i] = *reinterpret_cast<const bf16x8_t*>(Ahi + ao);
        al[i] = *reinterpret_cast<const bf16x8_t*>(Alo + ao);
        bh[i] = *reinterpret_cast<const bf16x8_t*>(Bhi + bo);
        bl[i] = *reinterpret_cast<const bf16x8_t*>(Blo + bo);
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
    if (EVT_GEMM_PRIO) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  int ncol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) ncol[j] = n0 + wn * 64 + j * 32 + lr;
  if (ksplit > 1) {
    // raw partial tile -> workspace plane `split`, compact row m; bias / act / scatter in splitk_finish_kernel
    float* wsp = g.ws + (int64_t)split * M * g.Nout;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (ncol[j] < g.Nout) wsp[(int64_t)m * g.Nout + ncol[j]] = acc[i][j][r];
      }
    return;
  }
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bv[j] = ncol[j] < g.Nout ? g.bias[ncol[j]] : 0.f;
  // The bias registers are pinned HERE (the empty asm reads them, so the load wait lands before the store loop).
  // With the load still pending inside the predicated store blocks the compiler re-waits `vmcnt(0)` in every block,
  // and on gfx9 vmcnt also counts stores -- every store then waited for the previous one to be acknowledged.
  asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t off = orow_off[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      if (off < 0) continue;
      float* orow = g.out + off;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (ncol[j] < g.Nout) {
          float v = acc[i][j][r] + bv[j];   // (activation per element here: doing all 64 up front costs 268 VGPRs -> 1 workgroup/CU)
          if (ACT == EVT_ACT_GELU_ERF) v = gelu_erf(v);
          orow[ncol[j]] = v;
        }
      }
    }
  }
}

int splitk_factor(int M, int K, int Nout) {
  return splitk_for(((M + 127) / 128) * ((Nout + 127) / 128), (K + 31) / 32);
}
// largest factor a launch with per-clip counts can ask for (one live row tile)
int splitk_factor_max(int K, int Nout) { return splitk_for((Nout + 127) / 128, (K + 31) / 32); }

// out[orow(m), n] = act(bias[n] + sum_s ws[s][m][n]), s ascending; one thread per 4 columns.
template <int ACT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const LinArgs g, int ksplit, int dyn, int tiles_n) {
  if (dyn) {
    ksplit = splitk_dynamic(g.count, g.B, tiles_n, (g.K + 31) / 32);
    if (ksplit == 1) return;  // the GEMM workgroups wrote `out` themselves
  }
  const int n4 = g.Nout >> 2;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int M = g.B * g.kcap;
  if (t >= (int64_t)M * n4) return;
  const int m = (int)(t / n4), c = (int)(t - (int64_t)m * n4) * 4;
  const int b = m / g.kcap, i = m - b * g.kcap;
  if (g.count != nullptr && i >= g.count[b]) return;
  const int64_t plane = (int64_t)M * g.Nout;
  const float* p = g.ws + (int64_t)m * g.Nout + c;
  float4 v = *reinterpret_cast<const float4*>(p);
  for (int s = 1; s < ksplit; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(p + s * plane);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  const float4 bv = *reinterpret_cast<const float4*>(g.bias + c);
  v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
  if (ACT == EVT_ACT_GELU_ERF) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
  const int64_t orow = (int64_t)b * g.o_rows + ((g.o_idx != nullptr) ? g.o_idx[m] : i);
  *reinterpret_cast<float4*>(g.out + orow * g.ldo + c) = v;
}

void launch_finish(const LinArgs& a, hipStream_t s, int ksplit, int dyn, int tiles_n) {
  const int64_t work = (int64_t)a.B * a.kcap * (a.Nout / 4);
  const dim3 fg((unsigned)((work + 255) / 256));
  if (a.act == EVT_ACT_GELU_ERF) hipLaunchKernelGGL(splitk_finish_kernel<EVT_ACT_GELU_ERF>, fg, dim3(256), 0, s, a, ksplit, dyn, tiles_n);
  else hipLaunchKernelGGL(splitk_finish_kernel<EVT_ACT_NONE>, fg, dim3(256), 0, s, a, ksplit, dyn, tiles_n);
}

template <int TBM, int TBN, int TBK, int WM, int WN>
void launch_split_cfg(const LinArgs& a, hipStream_t s, int ksplit = 1, int dyn = 0) {
  const int M = a.B * a.kcap;
  const int tiles_m = (M + TBM - 1) / TBM, tiles_n = (a.Nout + TBN - 1) / TBN;
  const dim3 grid(tiles_m * tiles_n * ksplit), block(WM * WN * 64);
  constexpr int TSP_ = (TBK == 32) ? TBK : TBK + 8;
  constexpr size_t lds_bytes = (size_t)(2 * TBM * TSP_ + 2 * TBN * TSP_) * 2 + (size_t)TBM * 8;
  EVT_ALLOW_LDS((gated_linear_split_kernel<EVT_ACT_GELU_ERF, TBM, TBN, TBK, WM, WN>), lds_bytes);
  EVT_ALLOW_LDS((gated_linear_split_kernel<EVT_ACT_NONE, TBM, TBN, TBK, WM, WN>), lds_bytes);
  const int tile_map = 0;   // (1: 4 x 2 tile blocks per XCD -- measured no better than the row-major walk)
  if (a.act == EVT_ACT_GELU_ERF)
    hipLaunchKernelGGL((gated_linear_split_kernel<EVT_ACT_GELU_ERF, TBM, TBN, TBK, WM, WN>), grid, block, lds_bytes, s, a,
                       tiles_n, tiles_m * tiles_n, tile_map, ksplit, dyn);
  else
    hipLaunchKernelGGL((gated_linear_split_kernel<EVT_ACT_NONE, TBM, TBN, TBK, WM, WN>), grid, block, lds_bytes, s, a,
                       tiles_n, tiles_m * tiles_n, tile_map, ksplit, dyn);
  if (ksplit > 1) launch_finish(a, s, ksplit, dyn, tiles_n);
}

void launch_split(const LinArgs& a, hipStream_t s) {
  if (evt_launch_split_big(a, s)) return;   // 256-row tiles when the launch fills the chip (evt_linear_pipe.hip)
  {   // a few hundred gated rows (one video stream): the latency-oriented kernel (evt_linear_small.hip)
    const int ks = evt_launch_split_small(a, s);
    if (ks > 1) launch_finish(a, s, ks, 0, (a.Nout + 127) / 128);
    if (ks >= 1) return;
  }
  if (a.ws != nullptr && (a.Nout & 3) == 0 && (a.ldo & 3) == 0) {   // split-K when the tile count leaves most CUs idle
    const int M = a.B * a.kcap;
    const bool dyn = a.count != nullptr && a.B <= EVT_SPLITK_DYN_MAX_CLIPS;
    const int ks = dyn ? splitk_factor_max(a.K, a.Nout) : splitk_factor(M, a.K, a.Nout);
    if (ks > 1 && (int64_t)ks * M * a.Nout * 4 <= a.ws_bytes) {
      launch_split_cfg<128, 128, 32, 2, 2>(a, s, ks, dyn ? 1 : 0);
      return;
    }
  }
  launch_split_cfg<128, 128, 32, 2, 2>(a, s);
}

// fp32 (rows, cols) -> hl32 planes (evt_linear.h): one thread per 4 consecutive columns of a row; groups past `cols`
// are zero-filled.
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out,
                                                            int64_t rows, int cols) {
  const int64_t pitch = hl32_pitch(cols);
  const int quads = (int)(pitch / 2 / 4);   // 4-column groups per (padded) row
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * quads) return;
  const int64_t r = t / quads;
  const int k = (int)(t - r * quads) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* src = w + r * cols + k;
  if (k + 4 <= cols && ((r * cols + k) & 3) == 0) v = *reinterpret_cast<const float4*>(src);
  else {
    if (k < cols) v.x = src[0];
    if (k + 1 < cols) v.y = src[1];
    if (k + 2 < cols) v.z = src[2];
    if (k + 3 < cols) v.w = src[3];
  }
  bf16x4_t h, l;
  split4(v, &h, &l);
  uint16_t* dst = out + r * pitch + hl32_hi(k);
  *reinterpret_cast<bf16x4_t*>(dst) = h;
  *reinterpret_cast<bf16x4_t*>(dst + 32) = l;
}

int launch_linear(const LinArgs& a, void* stream) {
  const int M = a.B * a.kcap;
  if (M == 0) return EVT_OK;
  const dim3 grid((a.Nout + BN - 1) / BN, (M + BM - 1) / BM), block(GEMM_THREADS);
  if (a.Wsplit != nullptr) {
    launch_split(a, evt_stream(stream));
  } else if (a.act == EVT_ACT_GELU_ERF)
    hipLaunchKernelGGL(gated_linear_kernel<EVT_ACT_GELU_ERF>, grid, block, 0, evt_stream(stream), a);
  else
    hipLaunchKernelGGL(gated_linear_kernel<EVT_ACT_NONE>, grid, block, 0, evt_stream(stream), a);
  return evt_check_launch("evt_gated_linear");
}

}  // namespace

extern "C" int evt_gated_linear(const evt_linear_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_gated_linear: null descriptor");
  EVT_REQUIRE(d->A && d->W && d->bias && d->out, EVT_ERR_BAD_ARG, "evt_gated_linear: null A/W/bias/out");
  EVT_REQUIRE(d->B >= 0 && d->kcap >= 0 && d->K > 0 && d->Nout > 0, EVT_ERR_BAD_ARG, "evt_gated_linear: bad sizes");
  EVT_REQUIRE((d->K & 3) == 0 && (d->lda & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_gated_linear: K=%d and lda must be multiples of 4", d->K);
  EVT_REQUIRE(d->lda >= d->K && d->ldo >= d->Nout, EVT_ERR_BAD_ARG, "evt_gated_linear: leading dimension too small");
  EVT_REQUIRE(d->act == EVT_ACT_NONE || d->act == EVT_ACT_GELU_ERF, EVT_ERR_BAD_ARG, "evt_gated_linear: act=%d", d->act);
  EVT_REQUIRE(d->p_upd == nullptr || d->a_idx != nullptr, EVT_ERR_BAD_ARG, "evt_gated_linear: p_upd needs a_idx");
  EVT_REQUIRE(d->a_rows > 0 && d->o_rows > 0, EVT_ERR_BAD_ARG, "evt_gated_linear: a_rows/o_rows must be positive");
  EVT_REQUIRE(d->W_split == nullptr || (d->K & 7) == 0, EVT_ERR_BAD_SHAPE, "evt_gated_linear: split weights need K %% 8 == 0 (K=%d)", d->K);
  EVT_REQUIRE(d->workspace_bytes >= 0, EVT_ERR_BAD_ARG, "evt_gated_linear: negative workspace_bytes");
  LinArgs a{d->A, d->lda, d->a_idx, d->a_rows, d->W, (const uint16_t*)d->W_split, d->bias, d->out, d->ldo, d->o_idx, d->o_rows,
            d->count, d->p_upd, d->B, d->kcap, d->K, d->Nout, d->act, (float*)d->workspace, d->workspace_bytes};
  if (d->a_bf16) {
    a.a_bf16 = 1;
    EVT_REQUIRE(d->act == EVT_ACT_NONE && evt_big_choice(a) != 0, EVT_ERR_BAD_SHAPE,
                "evt_gated_linear: bf16 activations only on the persistent 256-row kernel without activation (query "
                "evt_gated_linear_big_tile first): B*kcap=%d K=%d Nout=%d", d->B * d->kcap, d->K, d->Nout);
  }
  return launch_linear(a, stream);
}

extern "C" int evt_gated_linear_big_tile(const evt_linear_desc* d) {
  if (d == nullptr || d->B <= 0 || d->kcap <= 0 || d->K <= 0 || d->Nout <= 0) return 0;
  LinArgs a{d->A, d->lda, d->a_idx, d->a_rows, d->W, (const uint16_t*)d->W_split, d->bias, d->out, d->ldo, d->o_idx, d->o_rows,
            d->count, d->p_upd, d->B, d->kcap, d->K, d->Nout, d->act, (float*)d->workspace, d->workspace_bytes};
  return evt_big_choice(a);
}

extern "C" int64_t evt_gated_linear_workspace_bytes(int32_t B, int32_t kcap, int32_t K, int32_t Nout, int32_t has_count) {
  if (B <= 0 || kcap <= 0 || K <= 0 || Nout <= 0) return 0;
  const int64_t M = (int64_t)B * kcap;
  if (M > (1 << 20)) return 0;
  const bool dyn = has_count && B <= EVT_SPLITK_DYN_MAX_CLIPS;
  const int ks = dyn ? splitk_factor_max(K, Nout) : splitk_factor((int)M, K, Nout);
  return ks > 1 ? (int64_t)ks * M * Nout * 4 : 0;
}

extern "C" int evt_gated_mlp(const evt_mlp_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_gated_mlp: null descriptor");
  EVT_REQUIRE(d->A && d->W1 && d->b1 && d->W2 && d->b2 && d->hidden && d->out, EVT_ERR_BAD_ARG, "evt_gated_mlp: null pointer");
  EVT_REQUIRE(d->B >= 0 && d->kcap >= 0 && d->D > 0 && d->Dh > 0 && d->rows > 0, EVT_ERR_BAD_ARG, "evt_gated_mlp: bad sizes");
  EVT_REQUIRE((d->D & 3) == 0 && (d->Dh & 3) == 0 && (d->lda & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_gated_mlp: D, Dh, lda must be multiples of 4");
  EVT_REQUIRE(d->p_upd == nullptr || d->idx != nullptr, EVT_ERR_BAD_ARG, "evt_gated_mlp: p_upd needs idx");
  EVT_REQUIRE((d->W1_split == nullptr) == (d->W2_split == nullptr), EVT_ERR_BAD_ARG, "evt_gated_mlp: W1_split/W2_split must come together");
  EVT_REQUIRE(d->W1_split == nullptr || ((d->D & 7) == 0 && (d->Dh & 7) == 0), EVT_ERR_BAD_SHAPE, "evt_gated_mlp: split weights need D, Dh %% 8 == 0");
  LinArgs fc1{d->A, d->lda, d->idx, d->idx ? d->rows : d->kcap, d->W1, (const uint16_t*)d->W1_split, d->b1, d->hidden, (int64_t)d->Dh, nullptr,
              d->kcap, d->count, d->p_upd, d->B, d->kcap, d->D, d->Dh, EVT_ACT_GELU_ERF, (float*)d->workspace, d->workspace_bytes};
  LinArgs fc2{d->hidden, (int64_t)d->Dh, nullptr, d->kcap, d->W2, (const uint16_t*)d->W2_split, d->b2, d->out, d->ldo, d->idx,
              d->idx ? d->rows : d->kcap, d->count, nullptr, d->B, d->kcap, d->Dh, d->D, EVT_ACT_NONE, (float*)d->workspace, d->workspace_bytes};
  // Both launches on the 256-row kernel: the hidden scratch holds hl32 lines (same bytes as fp32) -- GELU(x) is split once,
  // in the first launch's epilogue, and the second launch stages it without conversion.
  if (fc1.Wsplit != nullptr && (d->Dh & 31) == 0 && evt_big_choice(fc1) != 0 && evt_big_choice(fc2) != 0) {
    fc1.out_planes = 1;
    fc2.a_planes = 1;
  }
  int rc = launch_linear(fc1, stream);
  if (rc != EVT_OK) return rc;
  return launch_linear(fc2, stream);
}

extern "C" int evt_split_weights(const float* w, void* out, int64_t rows, int64_t cols, void* stream) {
  EVT_REQUIRE(w != nullptr && out != nullptr && rows >= 0 && cols > 0, EVT_ERR_BAD_ARG, "evt_split_weights: bad arguments");
  EVT_REQUIRE(cols <= (1 << 24), EVT_ERR_BAD_SHAPE, "evt_split_weights: cols=%lld too large", (long long)cols);
  if (rows == 0) return EVT_OK;
  const int64_t work = rows * (hl32_pitch((int)cols) / 8);
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, evt_stream(stream), w,
                     (uint16_t*)out, rows, (int)cols);
  return evt_check_launch("evt_split_weights");
}

extern "C" int64_t evt_split_weights_bytes(int64_t rows, int64_t cols) {
  if (rows < 0 || cols <= 0) return 0;
  return rows * hl32_pitch((int)cols) * 2;
}
