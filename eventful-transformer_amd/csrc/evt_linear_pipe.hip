// evt_linear_pipe.hip -- K3/K7, split-precision gated linear for launches that fill the chip: persistent 256-row workgroup tiles
// with a software-pipelined k-tile (round 5; replaces the round-2 kernel evt_linear_big.hip, whose numbers are in profiles/r02-r04).
//
// Why tiles of 256 rows: a 128x128 tile moves 1024 operand bytes L2 -> CU per k for 32768 FLOP (32 FLOP/B, ~11 TB/s at 340 TF);
// 256x256 halves the bytes per FLOP (64 FLOP/B; 256x192: 55) and the per-element staging work.  ONE persistent workgroup per CU; the
// workgroups of an XCD walk a contiguous run of tiles side by side.  fp32 operands are bf16 hi + lo, three v_mfma_f32_32x32x16_bf16
// per product (lo.hi, hi.lo, hi.hi), fp32 accumulate; weights come pre-split as hl32 lines, activations are split while staged.
//
// What was wrong with the round-2 kernel (8 waves, 64x128 per wave, stage / multiply as two phases per k-tile, ONE fragment set):
// the matrix pipe idled during every fragment read, and the staging wave's conversions starved beside its partner's MFMA stream
// (0.47 matrix-pipe busy on QKV, 0.60 at best).  Here every wave runs ONE software-pipelined instruction stream per k-tile whose
// order is pinned in the source (a sched_barrier behind every "slot" = one MFMA + its share of the fillers):
//
//   slots 0 .. Z-1   : MFMAs of k-tile t (fragment set F0, then F1) | fillers: the fragment reads of its second k-half (-> F1),
//                      split + LDS stores of k-tile t+1 (registers R -> the other LDS stage), the global loads of k-tile t+2 into
//                      the registers just freed (a load has a whole k-tile to arrive)
//   ONE s_barrier per k-tile, at 3/4 of it (so the staging spreads over 3/4 of the MFMAs, ~2.5 fillers per MFMA)
//   slots Z .. end   : MFMAs on F1 | fillers: fragment reads of the first k-half of k-tile t+1 (-> F0), offset bumps
//
// so nothing in the stream waits for LDS or HBM except at the barrier.  Consecutive MFMAs write DIFFERENT accumulators (an instruction
// between two MFMAs on the same accumulator costs +43 cycles, MI355X_MICROARCH.md).  What was measured on the way (profiles/r05/,
// DESIGN.md section 6): one wave per SIMD (four 512-register waves, 128x128 per wave) is 3-10 % slower than two 256-register waves
// -- an in-order wave that is held up ISSUING a global load or an LDS store (~40 cycles each with four waves of a CU doing it at
// once) has nobody beside it to feed the matrix pipe; a load instruction must address whole 128-byte runs (8 lanes per row), not 64
// separate 16-byte pieces; 64-bit VALU address arithmetic is replaced by SGPR base + 32-bit running offsets; tile descriptions (gather
// indices -> byte offsets) are prepared two tiles ahead inside the previous epilogue and parked in LDS, so the k loop has no branchy
// bookkeeping; 256x192 tiles (QKV: 6.0 rounds of 256 CUs instead of 4.5 -> 5, no register spills) beat 256x256.
//
// MFMA operand roles: the weight fragment is the A operand, the activation fragment the B operand, so a lane of the accumulator holds
// ONE output row (token) and FOUR CONSECUTIVE output columns per register quad; the epilogue sends 32-row sub-tiles through a
// wave-private LDS buffer and stores whole contiguous 16-byte pieces of 8 rows per instruction (16 bytes per lane into 64 different
// 32-byte segments measured 29k cycles per tile; 4-byte stores, the round-2 form, 12k; this form ~9k for fp32 output).  Sums are formed
// in the same order as in the 128x128 and small-row kernels (k ascending; per accumulator lo.hi, hi.lo, hi.hi): results are
// bit-identical to theirs (tests/big_tile_check.py compares them).
//
// Formats (FMT bits): 1 = activations pre-split hl32 lines (the MLP's hidden scratch: staging is a copy), 2 = output written as
// hl32 lines (first half of evt_gated_mlp: GELU(x) is split once per element, in the epilogue), 4 = activations are ONE bf16
// plane of exactly bf16-representable values (the A.v state of a bf16 matmul_2_cast, blocks.py:183-189, which IS the attention
// output: hi = the value, lo = 0, the lo.hi MFMA is skipped, the gate reference is refreshed with the widened values).
// Gather through a_idx while staging, scatter through o_idx in the epilogue, gate reference refresh p[idx] = c[idx] (modules.py:151)
// from the staged fp32 registers, dealt over the column tiles.
#include "evt_linear.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#ifndef EVT_PIPE_ZBP     // barrier position inside a k-tile, in percent of a k-half past the middle (0: between the two k-halves)
#define EVT_PIPE_ZBP 50
#endif
#ifndef EVT_PIPE_RTAIL   // slots at the end of a k-tile that carry no fragment read (hipcc waits lgkmcnt(0) at the top of the next k-tile)
#define EVT_PIPE_RTAIL 4
#endif
#ifndef EVT_PIPE_STAIL   // slots in front of the barrier that carry no staging action (the barrier waits for the LDS stores)
#define EVT_PIPE_STAIL 3
#endif
#ifndef EVT_PIPE_FTAIL   // slots at the end of the first k-half that carry no fragment read
#define EVT_PIPE_FTAIL 8
#endif
#ifndef EVT_PIPE_PINEVERY   // a sched_barrier behind every n-th slot
#define EVT_PIPE_PINEVERY 1
#endif
#ifndef EVT_PIPE_PIN     // 1: sched_barrier between groups (the source order is the issue order)
#define EVT_PIPE_PIN 1
#endif

#define PIPE_PIN() do { if (EVT_PIPE_PIN) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef EVT_PROF   // phase timing of wave 0 of one workgroup (scripts/gemm_prof.py): s_memtime at the phase boundaries
__device__ unsigned long long evt_prof_pipe_buf[8];
#define PIPE_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define PIPE_TICK(slot) do { } while (0)
#endif

// f(integral_constant<int, I>) for I = B .. N-1, expanded at compile time (a `#pragma unroll` loop of 32 groups with this
// much code in it was left rolled: the accumulators, indexed by the loop variable, then lived in scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int ACT, int TBN, int FMT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void gated_linear_pipe_kernel(const LinArgs g, int tiles_n, int tiles_total) {
  // WAVES = 8 (shipped): two 256-register waves per SIMD, 64 x (TBN / 2) per wave, the same pipelined stream in each.  WAVES = 4 (kept
  // compilable, not instantiated): one 512-register wave per SIMD, 128 x (TBN / 2) per wave.
  constexpr int TBM = 256, TBK = 32, NT = WAVES * 64;
  static_assert(WAVES == 4 || WAVES == 8, "waves per workgroup");
  constexpr bool APL = (FMT & 1) != 0, OPL = (FMT & 2) != 0, ABF = (FMT & 4) != 0;
  static_assert(!(APL && ABF), "one activation format");
  constexpr int MI = 16 / WAVES, NJ = TBN / 64; // 32x32 accumulators per wave: NJ (weight rows) x MI (activation rows)
  constexpr int G = MI * NJ;                    // accumulators per wave
  static_assert(TBN % 64 == 0 && NJ >= 2 && NJ <= 4, "column tile");
  // ---- LDS image of one stage (bytes).  64-byte rows of 32 bf16, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3); the lo
  // planes start 64 bytes past a multiple of 128 where one ds_write_b128 lane group stores hi and lo chunks together.
  constexpr int APADB = APL ? 64 : 0, WPADB = 64;
  constexpr int A_HI = 0, A_LO = TBM * 64 + APADB, W_HI = A_LO + TBM * 64, W_LO = W_HI + TBN * 64 + WPADB, STAGEB = W_LO + TBN * 64;
  constexpr int AROWS = ABF ? NT / 4 : NT / 8;  // row distance of a thread's activation rows (bf16: 4 lanes per row, else 8)
  constexpr int WROWS = NT / 8;                 // ... of its weight rows
  constexpr int NAO = TBM / AROWS;              // activation rows per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char evt_gemm_pipe_smem[];
  unsigned char* const lds = evt_gemm_pipe_smem;
  uint32_t* const orow_tab = reinterpret_cast<uint32_t*>(lds + 2 * STAGEB);   // 3 x TBM: byte offset of each output row, ~0 = none
  float* const bias_tab = reinterpret_cast<float*>(orow_tab + 3 * TBM);       // 3 x TBN
  uint32_t* const stash = reinterpret_cast<uint32_t*>(bias_tab + 3 * TBN);    // NAO x NT: the next tile's activation row offsets of every thread

  // Persistent workgroups, XCD-aware (workgroup w runs on XCD w % 8): XCD x owns a contiguous run of row-major tiles and its
  // workgroups walk it side by side, so at any time one L2 serves neighbouring column tiles of a few row tiles.
  const int x8 = blockIdx.x % 8, c8 = blockIdx.x / 8;
  const int cx = gridDim.x / 8 + (x8 < (int)(gridDim.x % 8) ? 1 : 0);
  const int q8 = tiles_total / 8, r8 = tiles_total % 8;
  const int run0 = x8 * q8 + min(x8, r8), runlen = q8 + (x8 < r8 ? 1 : 0);
  if (c8 >= runlen) return;
  const int ntile = (runlen - c8 + cx - 1) / cx;
  const int nk = g.K / TBK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int M = g.B * g.kcap;
  const int lr = lane & 31, lh = lane >> 5;
#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == 8 && wave == 0;
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif

  // ---- staging roles: one 16-byte global load per (thread, row), and the lanes that share a row read CONSECUTIVE pieces of
  // it, so a load instruction asks for whole 128-byte (bf16: 64-byte) runs.  (The first version gave a thread 32 consecutive bytes
  // of an fp32 row as two loads: each instruction then addressed 64 separate 16-byte pieces, 32 bytes apart.)
  //   fp32 activations: thread = (row tid / 8 + 32 j, 4 floats tid % 8)  -> 8 bytes of the hi plane, 8 of the lo plane
  //   hl32 activations and weights: (row tid / 8 + 32 j, chunk tid % 8 of the 128-byte line);  bf16: (row tid / 4 + 64 j, chunk tid % 4)
  constexpr int NA = NAO;                       // 16-byte activation loads per thread and k-tile
  constexpr int NW = TBN / WROWS;               // weight rows (= 16-byte loads) per thread
  static_assert(TBN % WROWS == 0, "weight rows per thread");
  constexpr uint32_t ABYTES = ABF ? TBK * 2u : TBK * 4u;
  const int ar0 = ABF ? (tid >> 2) : (tid >> 3);
  const int achunk = ABF ? (tid & 3) : (tid & 7);
  const int wr0 = tid >> 3, wchunk = tid & 7;
  // LDS byte offset of this thread's piece of activation row ar0 (+ j * AROWS rows): hl32: plane by chunk; fp32: 8 bytes at k = 4 achunk
  const uint32_t a_st = (APL ? ((achunk & 4) ? A_LO : A_HI) + (((achunk & 3) ^ ((ar0 >> 2) & 3)) << 4)
                       : ABF ? A_HI + (((achunk & 3) ^ ((ar0 >> 2) & 3)) << 4)
                             : A_HI + ((((achunk >> 1) & 3) ^ ((ar0 >> 2) & 3)) << 4) + (achunk & 1) * 8) + ar0 * 64;
  const uint32_t w_st = ((wchunk & 4) ? W_LO : W_HI) + wr0 * 64 + (((wchunk & 3) ^ ((wr0 >> 2) & 3)) << 4);                    // + j * 32 rows
  const uint32_t fr0 = lr * 64 + (((0 + lh) ^ ((lr >> 2) & 3)) << 4), fr1 = lr * 64 + (((2 + lh) ^ ((lr >> 2) & 3)) << 4);
  const uint32_t fa = A_HI + wm * (MI * 32 * 64), fw = W_HI + wn * (NJ * 32 * 64);

  const int64_t wpitch = hl32_pitch(g.K);
  const bool do_upd = g.p_upd != nullptr && !APL;
  const char* const Abase = reinterpret_cast<const char*>(g.A);
  const char* const Wbase = reinterpret_cast<const char*>(g.Wsplit);
  char* const Pbase = reinterpret_cast<char*>(g.p_upd);
  auto tile_of = [&](int seq) __attribute__((always_inline)) { return run0 + c8 + seq * cx; };

  // ---- tile descriptions.  `prepare(seq)` turns the gate's index lists of tile `seq` into byte offsets: this thread's activation
  // rows go to its private LDS stash, the tile's output rows and bias values to slot seq % 3 of the tables.  It runs two tiles
  // ahead (inside the epilogue of tile seq - 2, whose stores hide the index loads); `take(seq)` -- LDS reads and a few multiplies,
  // no global load -- installs the description when the fetch side enters the tile, two k-tiles before the multiply side does.
  // (The first version prefetched the indices into registers inside the k loop: the merges of those registers cost register
  // moves and an lgkmcnt(0) at the top of EVERY k-tile.)
  struct Prep { int src[NAO], row; float bias; };
  auto prepare_issue = [&](int seq, Prep& P) __attribute__((always_inline)) {
    const int tile_i = tile_of(seq), bm_i = tile_i / tiles_n;
    const int m0 = bm_i * TBM;
    P.bias = g.bias[min((tile_i - bm_i * tiles_n) * TBN + (tid < TBN ? tid : 0), g.Nout - 1)];   // (threads past TBN / TBM: unused copies)
#pragma unroll
    for (int j = 0; j < NAO; ++j) {
      const int m = m0 + ar0 + AROWS * j;
      P.src[j] = (g.a_idx != nullptr) ? g.a_idx[m < M ? m : M - 1] : 0;
    }
    const int m = m0 + (tid < TBM ? tid : 0);
    P.row = (g.o_idx != nullptr) ? g.o_idx[m < M ? m : M - 1] : 0;
  };
  auto prepare_finish = [&](int seq, const Prep& P) __attribute__((always_inline)) {
    const int tile = tile_of(seq), bm = tile / tiles_n;
    const int m0 = bm * TBM, slot = seq % 3;
#pragma unroll
    for (int j = 0; j < NAO; ++j) {
      const int m = m0 + ar0 + AROWS * j;
      uint32_t o = 0;   // rows past M read row 0 and are never stored
      if (m < M) {
        const int b = m / g.kcap, i = m - b * g.kcap;
        o = (uint32_t)((b * g.a_rows + ((g.a_idx != nullptr) ? P.src[j] : i)) * (int)g.lda) * (ABF ? 2u : 4u);
      }
      stash[j * NT + tid] = o + (uint32_t)achunk * 16u;
    }
    const int m = m0 + tid;
    uint32_t off = ~0u;
    if (m < M) {
      const int b = m / g.kcap, i = m - b * g.kcap;
      off = (uint32_t)((b * g.o_rows + ((g.o_idx != nullptr) ? P.row : i)) * (int)g.ldo) * 4u;
    }
    if (tid < TBM) orow_tab[slot * TBM + tid] = off;
    if (tid < TBN) bias_tab[slot * TBN + tid] = P.bias;
  };
  // Byte offsets (from the scalar bases: global_load with an SGPR base and a 32-bit VGPR offset, no 64-bit address arithmetic) of
  // this thread's 16 bytes in its activation rows / weight rows, at the k-tile the NEXT loads fetch: they run along with the
  // fetch side (+ one k-tile behind every ktile(), in its idle second half) and are re-based by take().
  uint32_t a_off[NAO];
  uint32_t w_off[NW];
  int s_m0 = 0, s_upd = -1;    // that tile's first row; the next k-tile whose rows this column tile writes back to the gate reference
  auto take = [&](int seq) __attribute__((always_inline)) {
    const int tile = tile_of(seq), bm = tile / tiles_n, bn = tile - bm * tiles_n;
    s_m0 = bm * TBM;
    s_upd = do_upd ? bn : -1;
#pragma unroll
    for (int j = 0; j < NAO; ++j) a_off[j] = stash[j * NT + tid];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int n = bn * TBN + wr0 + WROWS * j;
      w_off[j] = (uint32_t)((n < g.Nout ? n : g.Nout - 1) * (int)wpitch) * 2u + (uint32_t)wchunk * 16u;
    }
  };

  // ---- registers of the pipeline
  f32x4 Ra[NA];                // raw activation data of stream element t + 1 (then t + 2)
  u32x4_t Rw[NW];              // its weight lines
  struct Frag { bf16x8_t ah[MI], al[MI], wh[NJ], wl[NJ]; };
  Frag F0, F1;
  f32x16 acc[NJ][MI];

  auto fetch_a = [&](int n) __attribute__((always_inline)) { Ra[n] = *reinterpret_cast<const f32x4*>(Abase + a_off[n]); };
  auto fetch_w = [&](int j) __attribute__((always_inline)) { Rw[j] = *reinterpret_cast<const u32x4_t*>(Wbase + w_off[j]); };
  auto bump = [&](int n) __attribute__((always_inline)) {   // n-th offset register -> the next k-tile
    if (n < NAO) a_off[n] += ABYTES; else w_off[n - NAO] += 128u;
  };
  auto store_w = [&](int j, uint32_t sb) __attribute__((always_inline)) {   // sb: byte offset of the LDS stage
    *reinterpret_cast<u32x4_t*>(lds + sb + w_st + j * (WROWS * 64)) = Rw[j];
  };
  // fp32 activations: a float4 (4 consecutive k of one row) -> 8 bytes of the hi plane, 8 of the lo plane
  auto split_pair = [&](float x, float y, uint32_t* h, uint32_t* l) __attribute__((always_inline)) {
    union { bf16x2_t b; uint32_t u; } hh, ll;
    hh.b = __builtin_convertvector((f32x2_t){x, y}, bf16x2_t);
    const float r0 = x - __uint_as_float(hh.u << 16), r1 = y - __uint_as_float(hh.u & 0xffff0000u);
    ll.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
    *h = hh.u;
    *l = ll.u;
  };
  u32x2_t Hq[NAO], Lq[NAO];        // fp32 activations: the split pieces between their conversion and their LDS store
  auto split_unit = [&](int u) __attribute__((always_inline)) {   // unit u = float pair (u & 1) of float4 (= row) u >> 1
    const f32x4 v = Ra[u >> 1];
    uint32_t h, l;
    if (u & 1) split_pair(v.z, v.w, &h, &l); else split_pair(v.x, v.y, &h, &l);
    Hq[u >> 1][u & 1] = h;
    Lq[u >> 1][u & 1] = l;
  };
  auto store_a = [&](int n, uint32_t sb) __attribute__((always_inline)) {   // row ar0 + AROWS n of the thread
    unsigned char* st = lds + sb + a_st + n * (AROWS * 64);
    if (APL || ABF) *reinterpret_cast<f32x4*>(st) = Ra[n];
    else {
      *reinterpret_cast<u32x2_t*>(st) = Hq[n];
      *reinterpret_cast<u32x2_t*>(st + (A_LO - A_HI)) = Lq[n];
    }
  };

  // The staging work of one k-tile as a list of actions, dealt over the slots in front of the barrier.
  //   fp32: per row n (NAO): U U A(n) [W]     (U = split unit: 6 VALU; A = two 8-byte LDS stores + the load of the next k-tile)
  //   hl32 / bf16: A(n) [W] ...            (A = one LDS store + one load)
  //   W(j) = one LDS store + one load
  struct Act { int kind, idx; };   // kind 0: split unit, 1: activation store + refill, 2: weight store + refill
  struct Sched {
    Act a[48];
    int n;
    constexpr Sched() : a(), n(0) {
      int w = 0;
      if (!APL && !ABF) {
        for (int p = 0; p < NAO; ++p) {
          a[n++] = {0, 2 * p}; a[n++] = {0, 2 * p + 1};
          a[n++] = {1, p};
          if (w < NW) a[n++] = {2, w++};
        }
      } else {
        for (int p = 0; p < NA; ++p) {
          a[n++] = {1, p};
          if (w < NW) a[n++] = {2, w++};
        }
      }
      while (w < NW) a[n++] = {2, w++};
    }
  };
  constexpr Sched SCH{};
  // One "slot" = one MFMA followed by its share of the fillers.  Slot m of a k-half: j = m / (P MI), pass = m / MI % P, i = m % MI
  // with P = 3 passes (lo.hi, hi.lo, hi.hi; 2 for bf16 activations) -- consecutive MFMAs write DIFFERENT accumulators and an
  // accumulator comes round again after MI = 4 of them.  Per accumulator the order of the three is that of the other kernels.
  constexpr int P = ABF ? 2 : 3, NS = P * G;     // slots per k-half
  constexpr int Z = NS + NS * EVT_PIPE_ZBP / 100;   // slots in front of the barrier
  constexpr int NR = 2 * (MI + NJ);              // fragment reads per k-half
  constexpr int ZS = Z - EVT_PIPE_STAIL;         // slots that carry staging actions
  constexpr int ZF = (NS - EVT_PIPE_FTAIL > NS / 2) ? NS - EVT_PIPE_FTAIL : NS / 2;   // slots that carry this k-tile's second fragment reads
  constexpr int ZR = (2 * NS - Z - EVT_PIPE_RTAIL > (2 * NS - Z) / 2) ? 2 * NS - Z - EVT_PIPE_RTAIL : (2 * NS - Z) / 2;
  static_assert(ZS >= 4 && ZR >= 2 && ZF >= 4 && ZS <= 2 * NS - (NAO + NW), "tails");

  // r-th fragment read of a k-half, in the order the slots consume them: wh0, al0..3, wl0, ah0..3, then wh_j, wl_j for j = 1 ..
  auto read_frag = [&](Frag& F, int r, uint32_t sb, uint32_t frk) __attribute__((always_inline)) {
    const unsigned char* st = lds + sb + frk;
    if (r == 0) F.wh[0] = *reinterpret_cast<const bf16x8_t*>(st + fw);
    else if (r <= MI) { if (!ABF) F.al[r - 1] = *reinterpret_cast<const bf16x8_t*>(st + fa + (A_LO - A_HI) + (r - 1) * (32 * 64)); }
    else if (r == MI + 1) F.wl[0] = *reinterpret_cast<const bf16x8_t*>(st + fw + (W_LO - W_HI));
    else if (r <= 2 * MI + 1) F.ah[r - MI - 2] = *reinterpret_cast<const bf16x8_t*>(st + fa + (r - MI - 2) * (32 * 64));
    else {
      const int j = 1 + ((r - 2 - 2 * MI) >> 1);
      if (((r - 2 - 2 * MI) & 1) == 0) F.wh[j] = *reinterpret_cast<const bf16x8_t*>(st + fw + j * (32 * 64));
      else F.wl[j] = *reinterpret_cast<const bf16x8_t*>(st + fw + (W_LO - W_HI) + j * (32 * 64));
    }
  };
  auto mfma_slot = [&](const Frag& F, int m) __attribute__((always_inline)) {
    const int j = m / (P * MI), pass = (m / MI) % P + (ABF ? 1 : 0), i = m % MI;
    if (pass == 0) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.wh[j], F.al[i], acc[j][i], 0, 0, 0);
    else if (pass == 1) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.wl[j], F.ah[i], acc[j][i], 0, 0, 0);
    else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.wh[j], F.ah[i], acc[j][i], 0, 0, 0);
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  uint32_t sb = 0;             // byte offset of the stage that holds the k-tile being multiplied

  // ---- epilogue.  The accumulator layout (lane = output row, register quad = 4 consecutive columns) would store 16 bytes per lane
  // into 64 different 32-byte segments per instruction -- measured at 29k cycles per 256x256 tile, a fifth of the launch.  The
  // sub-tile of 32 rows x CW columns therefore goes through a wave-private LDS buffer (in the stage that is free between two
  // tiles) and comes back with LPR consecutive lanes on one row: every store instruction writes whole contiguous 16-byte pieces
  // of RPI rows (fp32: 4 rows x 256 bytes).  No barrier inside: a wave's LDS operations execute in order.
  constexpr int JC = (NJ % 2 == 0 && WAVES == 4) ? 2 : 1, NC = NJ / JC, CW = JC * 32;   // column chunk of the transposition
  constexpr int PB = CW * 4 + 16;                                         // row pitch of the buffer: conflict-free 16-byte stores
  constexpr int VPL = OPL ? 8 : 4;                                        // values per lane and pass on the way out
  constexpr int LPR = CW / VPL, RPI = 64 / LPR, IT = 32 / RPI;            // lanes per row, rows per instruction, passes per chunk
  static_assert(WAVES * 32 * PB <= STAGEB, "transposition buffers fit the free stage");
  auto epilogue = [&](int seq) __attribute__((always_inline)) {
    const int tile = tile_of(seq), bm = tile / tiles_n, bn = tile - bm * tiles_n, slot = seq % 3;
    char* const obase = reinterpret_cast<char*>(g.out);
    unsigned char* const buf = lds + (STAGEB - sb) + wave * (32 * PB);
    const uint32_t* otab = orow_tab + slot * TBM + wm * (MI * 32) + lane / LPR;
    const int cq = (lane % LPR) * VPL;                                    // this lane's first column inside a chunk
    const int ncol0 = bn * TBN + wn * (NJ * 32) + cq;
    const float* btab = bias_tab + slot * TBN + wn * (NJ * 32) + cq;
    const bool interior = (bm + 1) * TBM <= M && (bn + 1) * TBN <= g.Nout;
    Prep prep;
    const bool more = seq + 2 < ntile;
    if (more) prepare_issue(seq + 2, prep);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int jj = 0; jj < JC; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x16& a = acc[c * JC + jj][i];
            const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            *reinterpret_cast<f32x4*>(buf + lr * PB + (jj * 32 + 8 * q + 4 * lh) * 4) = v;
          }
        f32x4 b0 = *reinterpret_cast<const f32x4*>(btab + c * CW), b1 = b0;
        if (OPL) b1 = *reinterpret_cast<const f32x4*>(btab + c * CW + 4);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const unsigned char* src = buf + (it * RPI + lane / LPR) * PB + cq * 4;
          const uint32_t off = otab[32 * i + it * RPI];
          const int ncol = ncol0 + c * CW;
          const bool pred = interior || (off != ~0u && ncol < g.Nout);   // (Nout % 4 == 0, hl32 output: % 32: a piece is whole or absent)
          f32x4 x0 = *reinterpret_cast<const f32x4*>(src);
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = x0[r] + b0[r];
          if (OPL) {
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 + r] = x1[r] + b1[r];
          }
          if (ACT == EVT_ACT_GELU_ERF) {
#pragma unroll
            for (int r = 0; r < VPL; r += 2) {
              const f32x2_t y = gelu_erf2((f32x2_t){v[r], v[r + 1]});
              v[r] = y.x;
              v[r + 1] = y.y;
            }
          }
          if (!OPL) {
            const f32x4 o = {v[0], v[1], v[2], v[3]};
            if (pred) *reinterpret_cast<f32x4*>(obase + (off + (uint32_t)ncol * 4u)) = o;
          } else {   // hl32 line of the 32-column group: 8 hi values at (n % 32) * 2, their lo values 64 bytes later
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));   // the ROUNDED values: no contraction into the residuals
            u32x4_t h, l;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              uint32_t hh, ll;
              split_pair(v[2 * r], v[2 * r + 1], &hh, &ll);
              h[r] = hh;
              l[r] = ll;
            }
            const uint32_t cb = (uint32_t)(ncol & ~31) * 4u + (uint32_t)(ncol & 31) * 2u;
            if (pred) {
              *reinterpret_cast<u32x4_t*>(obase + (off + cb)) = h;
              *reinterpret_cast<u32x4_t*>(obase + (off + cb + 64u)) = l;
            }
          }
        }
      }
    }
    if (more) prepare_finish(seq + 2, prep);
  };

  // ---- gate-reference refresh: the column tile bn writes the k-tiles bn, bn + tiles_n, ... of its rows back to p (from R, i.e.
  // from the fp32 values as loaded) before they are split
  auto refresh = [&](int k) __attribute__((always_inline)) {
    if (k != s_upd) return;
    s_upd += tiles_n;
#pragma unroll
    for (int j = 0; j < NAO; ++j) {
      if (s_m0 + ar0 + AROWS * j < M) {
        if (ABF) {   // widen the 8 bf16 values: the fp32 reference row has twice the byte offset
          union { f32x4 v; uint32_t u[4]; } in;
          in.v = Ra[j];
          f32x4 lo4, hi4;
          lo4.x = __uint_as_float(in.u[0] << 16); lo4.y = __uint_as_float(in.u[0] & 0xffff0000u);
          lo4.z = __uint_as_float(in.u[1] << 16); lo4.w = __uint_as_float(in.u[1] & 0xffff0000u);
          hi4.x = __uint_as_float(in.u[2] << 16); hi4.y = __uint_as_float(in.u[2] & 0xffff0000u);
          hi4.z = __uint_as_float(in.u[3] << 16); hi4.w = __uint_as_float(in.u[3] & 0xffff0000u);
          char* dst = Pbase + 2u * (a_off[j] - ABYTES);   // (a_off is one k-tile ahead of the element in R)
          *reinterpret_cast<f32x4*>(dst) = lo4;
          *reinterpret_cast<f32x4*>(dst + 16) = hi4;
        } else if (!APL) {
          *reinterpret_cast<f32x4*>(Pbase + (a_off[j] - ABYTES)) = Ra[j];
        }
      }
    }
  };

  // ---- one k-tile of the stream: multiplies the k-tile in stage sb (its first fragments are in F0), splits R into the other
  // stage, refills R from (Ak, Wk) and leaves the first fragments of the next k-tile in F0.  The call sites: the plain k loop of
  // a tile and the two k-tiles at its end, in which the fetch side is already in the next tile.
  auto ktile = [&]() __attribute__((always_inline)) {
    const uint32_t so = STAGEB - sb;   // the other stage
    PIPE_TICK(0);   // refresh + fetch-side bookkeeping
#ifdef EVT_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PIPE_TICK(1);   // waiting for the loads of the k-tile about to be staged (and the fragment reads)
#endif
    PIPE_PIN();
    static_for<0, 2 * NS>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m == NS) PIPE_TICK(2);
      if constexpr (m == Z) {
        barrier();
        PIPE_TICK(3);
        PIPE_PIN();
      }
      if constexpr (m < NS) mfma_slot(F0, m); else mfma_slot(F1, m - NS);
      // fragment reads: k-half 1 of this k-tile during its first half; k-half 0 of the next k-tile behind the barrier
      if constexpr (m < ZF) {
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value;
          if constexpr (r * ZF / NR == m) read_frag(F1, r, sb, fr1);
        });
      }
      if constexpr (m >= Z) {
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value;
          if constexpr (r * ZR / NR == m - Z) read_frag(F0, r, so, fr0);
        });
      }
      if constexpr (m < ZS) {
        static_for<0, SCH.n>([&](auto nc) __attribute__((always_inline)) {
          constexpr int n = decltype(nc)::value;
          if constexpr (n * ZS / SCH.n == m) {
            constexpr int kind = SCH.a[n].kind, idx = SCH.a[n].idx;
            if constexpr (kind == 0) split_unit(idx);
            else if constexpr (kind == 1) {
              store_a(idx, so);
              fetch_a(idx);
            } else {
              store_w(idx, so);
              fetch_w(idx);
            }
          }
        });
      }
      if constexpr (m >= 2 * NS - (NAO + NW)) bump(m - (2 * NS - (NAO + NW)));   // (behind the last load of this k-tile: ZS <= 2 NS - NAO - NW)
      if constexpr ((m % EVT_PIPE_PINEVERY) == EVT_PIPE_PINEVERY - 1) PIPE_PIN();
    });
    PIPE_TICK(4);   // second k-half
    sb = so;
  };

  // ---- prologue: descriptions of tiles 0 and 1, element (0, 0) -> stage 0, element (0, 1) -> R, F0 = k-half 0 of element (0, 0)
  {
    Prep p0;
    prepare_issue(0, p0);
    prepare_finish(0, p0);
    take(0);
    if (ntile > 1) {
      Prep p1;
      prepare_issue(1, p1);
      prepare_finish(1, p1);   // (the stash is this thread's own: take(0) has read it)
    }
#pragma unroll
    for (int n = 0; n < NA; ++n) fetch_a(n);
#pragma unroll
    for (int j = 0; j < NW; ++j) fetch_w(j);
#pragma unroll
    for (int n = 0; n < NAO + NW; ++n) bump(n);
    refresh(0);
    if (!APL && !ABF) {
#pragma unroll
      for (int u = 0; u < 2 * NAO; ++u) split_unit(u);
    }
#pragma unroll
    for (int n = 0; n < NA; ++n) store_a(n, 0u);
#pragma unroll
    for (int j = 0; j < NW; ++j) store_w(j, 0u);
#pragma unroll
    for (int n = 0; n < NA; ++n) fetch_a(n);
#pragma unroll
    for (int j = 0; j < NW; ++j) fetch_w(j);
#pragma unroll
    for (int n = 0; n < NAO + NW; ++n) bump(n);
    barrier();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_frag(F0, r, 0u, fr0);
  }

  // ---- tiles x k-tiles.  In iteration kt of tile seq the multiply side is at (seq, kt), R holds (seq, kt + 1) and the loads
  // of (seq, kt + 2) are issued; the last two k-tiles fetch (seq + 1, 0) and (seq + 1, 1).
  for (int seq = 0; seq < ntile; ++seq) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    for (int kt = 0; kt < nk - 2; ++kt) {
      refresh(kt + 1);
      ktile();
    }
    refresh(nk - 1);
    if (seq + 1 < ntile) take(seq + 1);
    else {                                 // past the last tile the stream re-reads the first k-tiles of this one (and stages them where nobody reads)
      s_upd = -1;
#pragma unroll
      for (int j = 0; j < NAO; ++j) a_off[j] -= (uint32_t)nk * ABYTES;
#pragma unroll
      for (int j = 0; j < NW; ++j) w_off[j] -= (uint32_t)nk * 128u;
    }
    ktile();
    refresh(0);
    ktile();
    epilogue(seq);
    PIPE_TICK(5);
    barrier();                             // the next k-tile's staging overwrites the stage the transposition buffers live in
  }
#ifdef EVT_PROF
  if (prof_on && lane == 0)
    for (int q = 0; q < 8; ++q) evt_prof_pipe_buf[q] = prof_acc[q];
#endif
}

template <int ACT, int TBN, int FMT, int WAVES>
void launch_pipe_waves(const LinArgs& a, hipStream_t s, dim3 grid, int tiles_n, int tiles_total) {
  constexpr size_t stage = (size_t)256 * 64 * 2 + ((FMT & 1) ? 64 : 0) + (size_t)TBN * 64 * 2 + 64;
  constexpr size_t lds_bytes = 2 * stage + (size_t)3 * 256 * 4 + (size_t)3 * TBN * 4 + (size_t)8 * 256 * 4;
  EVT_ALLOW_LDS((gated_linear_pipe_kernel<ACT, TBN, FMT, WAVES>), lds_bytes);
  hipLaunchKernelGGL((gated_linear_pipe_kernel<ACT, TBN, FMT, WAVES>), grid, dim3(WAVES * 64), lds_bytes, s, a, tiles_n, tiles_total);
}

// Eight waves.  (The kernel is written for 4 or 8; four 512-register waves -- one per SIMD, 128 x 128 per wave, half the LDS fragment
// traffic -- measured 3-10 % slower on every headline launch: an in-order wave that is held up issuing a load or an LDS store has
// nobody beside it to feed the matrix pipe.  profiles/r05/gemm_tile_shapes.txt.)
template <int ACT, int TBN, int FMT>
void launch_pipe_one(const LinArgs& a, hipStream_t s, dim3 grid, int tiles_n, int tiles_total) {
  launch_pipe_waves<ACT, TBN, FMT, 8>(a, s, grid, tiles_n, tiles_total);
}

template <int TBN>
void launch_pipe_cfg(const LinArgs& a, hipStream_t s) {
  const int M = a.B * a.kcap;
  const int tiles_m = (M + 255) / 256, tiles_n = (a.Nout + TBN - 1) / TBN;
  const int tt = tiles_m * tiles_n;
  const dim3 grid(std::min(tt, evt_cu_count()));
  if (a.a_bf16) launch_pipe_one<EVT_ACT_NONE, TBN, 4>(a, s, grid, tiles_n, tt);
  else if (a.a_planes) launch_pipe_one<EVT_ACT_NONE, TBN, 1>(a, s, grid, tiles_n, tt);
  else if (a.out_planes) launch_pipe_one<EVT_ACT_GELU_ERF, TBN, 2>(a, s, grid, tiles_n, tt);
  else if (a.act == EVT_ACT_GELU_ERF) launch_pipe_one<EVT_ACT_GELU_ERF, TBN, 0>(a, s, grid, tiles_n, tt);
  else launch_pipe_one<EVT_ACT_NONE, TBN, 0>(a, s, grid, tiles_n, tt);
}

}  // namespace

// Picks a 256-row tile when the launch has enough of them to fill the chip; 0 when the 128x128 kernel (or its split-K form, or the
// small-row-count kernel) should run instead.  EVT_GEMM_BIG: 0 never, 1 (default) automatic, 2 always 256x256, 3 always 256x128,
// 4 always 256x192 (tests/test_gpu_big_tiles.py forces each one).
int evt_big_choice(const LinArgs& a) {
  static const int mode = getenv("EVT_GEMM_BIG") ? atoi(getenv("EVT_GEMM_BIG")) : 1;
  // whole 32-k tiles, at least two of them; top-k gating only (the threshold policy's masked rows stay with the 128x128 kernel,
  // which skips dead tiles); 16-byte output pieces
  if (mode == 0 || a.Wsplit == nullptr || (a.K & 31) != 0 || a.K < 64 || a.count != nullptr || (a.Nout & 3) != 0 || (a.ldo & 3) != 0) return 0;
  // 32-bit byte offsets inside the kernel: activations (and the gate reference, same shape), weight planes and output below 4 GB
  // (the activation bound is taken at 4 bytes per element also for a bf16 launch: it then covers the fp32 gate reference)
  if ((int64_t)a.B * a.a_rows * a.lda * 4 >= ((int64_t)1 << 32) ||
      (int64_t)a.Nout * hl32_pitch(a.K) * 2 >= ((int64_t)1 << 32) ||
      (int64_t)a.B * a.o_rows * a.ldo * 4 >= ((int64_t)1 << 32))
    return 0;
  if (mode >= 2 && mode <= 4) return mode;
  // One persistent workgroup per CU: a launch of T tiles runs in ceil(T / CUs) rounds.  Take a tile whose columns divide Nout (no
  // wasted edge columns) and whose rounds are at least 75 % full on average (the dense first frame of B = 256 clips: projection / MLP-2 are
  // 788 tiles of 256x192 = 3.08 rounds -- 77 % -- and still beat the 128x128 kernel, which took them until round 5).  256x192 first: QKV (Nout = 2304) is 1536 tiles = 6.0 rounds
  // instead of the 4.5 (-> 5) of 256x256, and the 192-column instantiations keep every register: 310 vs 346 us for QKV, 441 vs 497
  // for MLP-1 + GELU, 790 vs 859 for the MLP pair at B = 256 (profiles/r05/gemm_tile_shapes.txt).
  const int cus = evt_cu_count(), M = a.B * a.kcap;
  const int tiles_m = (M + 255) / 256;
  auto fills = [&](int tbn) {
    if (a.Nout % tbn != 0) return false;
    const int tiles = tiles_m * (a.Nout / tbn), rounds = (tiles + cus - 1) / cus;
    return tiles >= cus && tiles * 100 >= rounds * cus * 75;
  };
  if (fills(192)) return 4;
  if (fills(256)) return 2;
  return 0;
}

bool evt_launch_split_big(const LinArgs& a, hipStream_t s) {
  switch (evt_big_choice(a)) {
    case 2: launch_pipe_cfg<256>(a, s); return true;
    case 3: launch_pipe_cfg<128>(a, s); return true;
    case 4: launch_pipe_cfg<192>(a, s); return true;
    default: return false;
  }
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof_pipe(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_pipe_buf), sizeof(unsigned long long) * 8);
}
#endif
