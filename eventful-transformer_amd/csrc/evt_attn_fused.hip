// evt_attn_fused.hip -- K5+K6 fused for the gated frames of EventfulBlock:
//
//   row softmax statistics over the q.k^T state (+ rel-pos terms)             blocks.py:518-522
//   -> gather the k selected columns, A delta gate (a~, da~, reference update) modules.py:187-201 ("col")
//   -> state[i,:] += round(a~ . dv~) ; += round(da~ . v_old), heads merged     modules.py:285-295
//
// One workgroup owns 32 attention rows of one (clip, head).  The N x N state row is streamed from
// HBM for the max / sum statistics only (never staged whole), the k gathered probabilities go
// straight into LDS tiles that feed the matrix cores, so a~ / da~ (H*N*k per clip) never touch
// HBM.  The selected-column axis is processed in chunks of 128, so any k (incl. the threshold
// policy's device-side count up to N) fits in the same 60 KB of LDS.
//
// Arithmetic per store type T (= the reference's `matmul_2_cast` dtype):
//   bf16 / fp16 : operands ARE bf16/fp16 values (the reference rounded them), so the products are
//                 exact on v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate -- what a bf16 matmul is;
//   fp32        : v_mfma_f32_32x32x2_f32.
// Every rounding point of the reference is kept: a~, da~ = round(a~ - ref), each matmul result, each +=.
//
// v operands come TRANSPOSED from K6a (evt_v_gate transposed=1): (B, D, kcap), k contiguous, so both
// MFMA operands are staged with 16-byte copies and read from LDS as 16-byte k-contiguous fragments.
// For N <= 256 the exponentials of the workgroup's 32 rows live in an LDS tile, so the gather of the
// selected columns is an LDS lookup; larger N re-reads the (L2-hot) state row and recomputes k exps.
//
// QK mode (product == nullptr; N == Nk <= 256, head dim 64): the score rows are not read from the q.k^T state at
// all but computed in the kernel from the CURRENT token buffer -- (q / scale) k^T on the matrix cores, q and k read
// from the buffer straight into MFMA fragments, into the same LDS tile.  At the ViViT operating point (k / N = 0.65) the state's row +
// column panel update (K4) touches 88 % of the state and costs more than this full recompute, and the state would
// only be written (K4) to be read once here: K4 and 2 x 477 MB of state traffic per launch at B = 256 go away.
#include "evt_attn_tiles.h"   // Tile<T> (LDS pitch + MFMA sweep), fast_exp, split4
#include "evt_prep_roles.h"
#include <algorithm>

namespace {

#ifdef EVT_PROF   // phase timing of wave 0 of one workgroup (scripts/attn_prof.py)
__device__ unsigned long long evt_prof_attn_buf[16];
#define ATT_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define ATT_TICK(slot) do { } while (0)
#endif

struct FusedArgs {
  const float* product; const float* qkv; const float* rel_y; const float* rel_x;
  void* a_state; const int32_t* idx; const int32_t* count;
  const void* v_delta_t; const void* v_old_t; void* pv; float* out_f32;
  int B, H, N, Nk, D, dh, kcap, gh, gw, qw;   // N rows x Nk columns; gh x gw: KEY grid; qw: query grid width
  float scale;                                // QK mode: q / scale (blocks.py:514)
  int qk_split;                               // QK mode: 1 = bf16 hi/lo split products, 0 = exact fp32 products
  const float* norm_ref; float* norm_parts;   // optional: (B,N,D) reference of the next gate -> (B,N,H) partial ||out - ref||^2
  const float* rel_terms;                     // optional: (B,H,N,gh+gw) rel-pos dot products from evt_rel_terms
};

constexpr int QKC = 64;       // QK mode: keys per chunk (16 per wave)
// Row pitch (floats) of the score / exp tile: 4 x an odd number -- rows are 16-byte aligned (the q.k^T phase stores 4 consecutive
// keys of a row as one ds_write_b128) and 16 consecutive rows start 4 (mod 64) x odd banks apart: those stores are conflict-free.
__host__ __device__ inline int fused_ep(int Nk) { return 4 * (((Nk + 3) >> 2) | 1); }
typedef float f32x4_acc __attribute__((ext_vector_type(4)));

// TPW = 32-column tiles per wave = dh / 64.  NREG > 0: N <= 64*NREG and the 8 rows a wave owns are held
// in registers (one HBM pass, all 8*NREG loads in flight together); NREG == 0: any N, two streamed passes.
// QK: the score tile is computed here (needs NREG > 0, TPW == 1) instead of being read from the state:
// 1 = exact fp32 products (v_mfma_f32_16x16x4_f32), 2 = split precision (q, k as bf16 hi + lo, three
// v_mfma_f32_16x16x32_bf16 per product, ~1e-5 relative -- the arithmetic K4 uses by default; 5x less matrix-pipe time).
// bf16 / fp16 store at head dim 64 (ViViT / ViT-B with matmul_2_cast): three workgroups per CU, i.e. at most 168 registers per
// lane incl. AGPRs -- at 170 the kernel dropped to two and the gated ViViT launch went from 480 to 610 us.  The fp32-store and
// head-dim-128 variants, and the streamed any-N variants (two row buffers in flight), need more registers than that (they
// would spill) and stay unconstrained.
template <typename T, int TPW, int NREG, int QK = 0>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && TPW == 1 && NREG > 0) ? 3 : 1) void softmax_av_gated_kernel(const FusedArgs a) {
  constexpr int P = Tile<T>::PITCH;
  constexpr int DHC = 64 * TPW;                        // head dim of this instantiation (== a.dh: launch_fused)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* An = reinterpret_cast<T*>(smem_raw);              // [FR][P]
  T* Ad = An + FR * P;                                 // [FR][P]
  T* Vd = Ad + FR * P;                                 // [dh][P]
  T* Vo = Vd + DHC * P;                                // [dh][P]
  float* relv = reinterpret_cast<float*>(Vo + DHC * P);   // [FR][gh+gw] rel-pos terms per row
  float* et = relv + FR * (a.gh + a.gw);               // [FR][EP] exp(x - max) (NREG > 0 only)
  // The V tiles are idle before the chunk loop and after it; they double as
  float* qs = reinterpret_cast<float*>(Vd);            // [FR][dh] q rows for the rel-pos dots (prologue)
  float* red1 = reinterpret_cast<float*>(Vd);          // [FR][dh] round(a~ . dv~)          (epilogue)
  constexpr int RDP = DHC + 4;                         // row pitch of the two tiles: rows land 4 banks apart (16-byte reads of
                                                       // neighbouring rows do not collide); 2 FR RDP floats <= the V tiles
  static_assert((size_t)2 * FR * RDP * sizeof(float) <= (size_t)2 * DHC * P * sizeof(T), "the rounded-product tiles alias the V tiles");
  float* red2 = red1 + FR * RDP;                       // [FR][RDP] round(da~ . v_old)
  const int EP = fused_ep(a.Nk);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the same value in a scalar register: row indices of a wave are wave-uniform
  // XCD-aware placement: workgroups are dispatched round-robin over the 8 XCDs in linear order (x fastest), which
  // would spread the row tiles of one (clip, head) over 8 private L2s.  Remapped so that all row tiles of a head run on
  // ONE XCD: its dv~ / v_old tiles -- and, in QK mode, its K rows -- are fetched into one L2 instead of up to eight.
  int bh = blockIdx.y, tile_x = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const int p = blockIdx.x + gridDim.x * blockIdx.y, x = p & 7, s = p >> 3, hl = s / (int)gridDim.x;
    tile_x = s - hl * (int)gridDim.x;
    bh = hl * 8 + x;
  }
  const int b = bh / a.H, h = bh - b * a.H;
  const int i0 = tile_x * FR;
#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == 2 && blockIdx.y == (gridDim.y > 1000 ? 1000u : gridDim.y / 2) && wave == 0;
  unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif
  const int cnt = a.count ? a.count[b] : a.kcap;
  const bool rel = a.rel_y != nullptr;
  const int nrel = a.gh + a.gw;
  const float inv_gw = rel ? 1.0f / (float)a.gw : 0.f;
  const float* prod = a.product + (int64_t)bh * a.N * a.Nk;
  T* st = reinterpret_cast<T*>(a.a_state) + (int64_t)bh * a.N * a.Nk;
  const int32_t* ix = a.idx + (int64_t)b * a.kcap;
  // Addressing.  Every tensor is reached through a wave-uniform base (scalar registers: the (clip, head) / clip slice) plus a
  // 32-bit BYTE offset -- one integer add per access instead of a 64-bit multiply-add chain per lane (the launch is bound by the
  // vector ALU: 1 894 VALU instructions per wave at ~70 % VALU busy, profiles/r05/attn_pmc_counters.txt).  Every slice addressed this
  // way is below 4 GB: evt_softmax_av_gated checks it (EVT_ERR_BAD_SHAPE otherwise).
  char* const stb = reinterpret_cast<char*>(st);
  auto st_off = [&](int row, int col) __attribute__((always_inline)) { return (uint32_t)(row * a.Nk + col) * (uint32_t)sizeof(T); };
  auto st_load = [&](uint32_t off) __attribute__((always_inline)) { return Store<T>::load(reinterpret_cast<const T*>(stb + off)); };
  auto st_store = [&](uint32_t off, float v) __attribute__((always_inline)) { Store<T>::store(reinterpret_cast<T*>(stb + off), v); };

  // Everything the chunk loop and the epilogue read from HBM that does not depend on the softmax is requested up
  // front (registers), so one workgroup waits for ~2 dependent round trips instead of one per phase: the index list
  // first (tiny), then -- behind the row loads of phase 1 -- the old a~ values of the first PF chunks, the first
  // dv~ / v_old chunk and the A.v state rows of the epilogue.
  constexpr int PF = (NREG > 0) ? 2 : 0;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int VIT = 64 * TPW * (FKC / VEC) / 256;   // 16-byte V pieces per thread per chunk
  int jpf[PF > 0 ? PF : 1];
  float oldpf[PF > 0 ? PF : 1][8];
#pragma unroll
  for (int c = 0; c < PF; ++c) {   // PF > 0 is only launched with kcap > 0: ix[0] is readable
    const int kk = c * FKC + lane;
    const int j = ix[kk < cnt ? kk : 0];
    jpf[c] = (kk < cnt) ? j : -1;
  }
  const T* Vg_d = reinterpret_cast<const T*>(a.v_delta_t) + (int64_t)bh * a.dh * a.kcap;
  const T* Vg_o = reinterpret_cast<const T*>(a.v_old_t) + (int64_t)bh * a.dh * a.kcap;
  const bool vvec = a.kcap > 0 && (a.kcap % VEC) == 0;
  uint4 vpd[VIT], vpo[VIT];
  uint4 vpd1[VIT], vpo1[VIT];   // PF path: the second chunk's pieces, requested before the first chunk is processed
  auto load_v_into = [&](int k0, uint4* pd, uint4* po) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC, kk = k0 + jj;
      const bool in = kk < a.kcap;
      const uint32_t o = (uint32_t)(d * a.kcap + (in ? kk : 0)) * (uint32_t)sizeof(T);
      const uint4 xd = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Vg_d) + o), xo = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Vg_o) + o);
      pd[it] = in ? xd : make_uint4(0, 0, 0, 0);
      po[it] = in ? xo : make_uint4(0, 0, 0, 0);
    }
  };
  auto load_v = [&](int k0) __attribute__((always_inline)) {
    // chunk k0 of dv~^T / v_old^T (k contiguous) -> registers.  Columns in [count, kcap) hold zeros (evt_v_gate
    // writes them) and meet a~ = da~ = 0 anyway; pieces past kcap are zeroed here.  Branch-free (clamped address +
    // select) so that the loads stay where they are issued instead of being sunk next to their use.
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC, kk = k0 + jj;
      const bool in = kk < a.kcap;   // kcap % VEC == 0: a piece is wholly in or out
      const uint32_t o = (uint32_t)(d * a.kcap + (in ? kk : 0)) * (uint32_t)sizeof(T);
      const uint4 xd = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Vg_d) + o), xo = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Vg_o) + o);
      vpd[it] = in ? xd : make_uint4(0, 0, 0, 0);
      vpo[it] = in ? xo : make_uint4(0, 0, 0, 0);
    }
  };
  T* pv = reinterpret_cast<T*>(a.pv);
  T* const pv_bh = pv + ((int64_t)b * a.N * a.D + h * DHC);                                   // this clip's rows, this head's channels
  const float* const nref_bh = a.norm_ref ? a.norm_ref + ((int64_t)b * a.N * a.D + h * DHC) : nullptr;
  constexpr int PIT = FR * (64 * TPW / 8) / 256;      // 8-channel state pieces per thread (epilogue)
  union Pv8 { uint4 u[(8 * sizeof(T)) / 16]; T t[8]; };
  Pv8 pvr[PIT];
  union Ref8 { float4 v[2]; float f[8]; };
  Ref8 nrr[PIT];   // the next gate's reference values of the same 8 channels (norm_ref), for the fused delta norm
  auto load_pv = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + 256 * it, row = e / (DHC / 8), c8 = (e - row * (DHC / 8)) * 8, i = i0 + row;
      const uint32_t o = (uint32_t)((i < a.N ? i : a.N - 1) * a.D + c8);   // element offset; clamped: branch-free
#pragma unroll
      for (int q = 0; q < (int)((8 * sizeof(T)) / 16); ++q)
        pvr[it].u[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pv_bh) + (o * (uint32_t)sizeof(T) + 16u * q));
      if (a.norm_ref != nullptr) {   // wave-uniform
        nrr[it].v[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nref_bh) + o * 4u);
        nrr[it].v[1] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nref_bh) + (o * 4u + 16u));
      }
    }
  };

  // ---- phase 1: per-row softmax statistics; wave w owns rows w*8 .. w*8+7 -----------------------
  float rmax[8], rsum[8];
  if (rel && a.rel_terms != nullptr) {
    // rel-pos terms computed once per frame by evt_rel_terms: copy the tile's 32 x (gh + gw) block (contiguous rows).
    // (Computed here, a workgroup reads 32 x (gh + gw) table rows of 256 bytes -- 688 KB at 42 x 42, 38 % of its life.)
    const float* src = a.rel_terms + ((int64_t)bh * a.N + i0) * nrel;
    const int rows = min(FR, a.N - i0);
    for (int e = tid; e < rows * nrel; e += 256) relv[e] = src[e];
    __syncthreads();
  } else if (rel) {
    // rel-pos terms of the workgroup's 32 rows (utils.py:159-168): q rows staged once in LDS, then the
    // 32 x (gh + gw) dot products are spread over all 256 threads (8 threads per row), head dim unrolled
    // so the 16-byte table loads of one dot are all in flight together.
    constexpr int DH = 64 * TPW;
    for (int e = tid; e < FR * (DH / 4); e += 256) {
      const int r = e / (DH / 4), c4 = e - r * (DH / 4), i = i0 + r;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < a.N) q = *reinterpret_cast<const float4*>(a.qkv + ((int64_t)b * a.N + i) * 3 * a.D + h * DH + c4 * 4);
      *reinterpret_cast<float4*>(qs + r * DH + c4 * 4) = q;
    }
    __syncthreads();
    {
      const int r = tid >> 3, sub = tid & 7, i = i0 + r;
      if (i < a.N) {
        const int yi = i / a.qw, xi = i - yi * a.qw;
        const float* q = qs + r * DH;
        for (int e = sub; e < nrel; e += 8) {
          const float* tab = (e < a.gh) ? a.rel_y + ((int64_t)yi * a.gh + e) * DH
                                        : a.rel_x + ((int64_t)xi * a.gw + (e - a.gh)) * DH;
          float4 t[DH / 4];
#pragma unroll
          for (int d = 0; d < DH / 4; ++d) t[d] = *reinterpret_cast<const float4*>(tab + d * 4);
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int d = 0; d < DH / 4; d += 2) {
            s0 += q[4 * d] * t[d].x + q[4 * d + 1] * t[d].y + q[4 * d + 2] * t[d].z + q[4 * d + 3] * t[d].w;
            s1 += q[4 * d + 4] * t[d + 1].x + q[4 * d + 5] * t[d + 1].y + q[4 * d + 6] * t[d + 1].z + q[4 * d + 7] * t[d + 1].w;
          }
          relv[r * nrel + e] = s0 + s1;
        }
      }
    }
    __syncthreads();
  }
  // Requests for everything the chunk loop and the epilogue need that does not depend on the scores: old a~ values of
  // the first PF chunks, the first dv~ / v_old chunk, the A.v state rows.  Issued behind the state-row loads of phase 1
  // or, in QK mode, ahead of the q.k^T phase, so that one workgroup waits for ~2 dependent round trips, not one per phase.
  auto prefetch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < PF; ++c)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int i = i0 + wave_s * 8 + rr;   // (scalar)
        oldpf[c][rr] = st_load(st_off(i < a.N ? i : 0, jpf[c] >= 0 ? jpf[c] : 0));   // branch-free
      }
    if (vvec) load_v(0);
    load_pv();
  };
  if (QK && NREG > 0) {
    // ---- phase 0 (QK mode): S = (q / scale) k^T for the 32 rows into the LDS tile `et` (pitch EP).  Wave w owns keys
    // 16w .. 16w+15 of every 64-key chunk; lane (l15, kg) holds, of q row l15 / key l15, the channels `chan` of its
    // fragment pieces ("Operand path" below); one workgroup barrier (the q rows), none per chunk.
    const int rs = 3 * a.D;   // floats per token row
    const float* clip = a.qkv + (int64_t)b * a.N * rs;
    const char* const qbase = reinterpret_cast<const char*>(clip + h * 64);           // this head's q channels of token 0
    const char* const kbase_b = reinterpret_cast<const char*>(clip + a.D + h * 64);   // ... its k channels
    const int l15 = lane & 15, kg = lane >> 4;
    const float inv = 1.0f / a.scale;
    const bool pow2 = (inv * a.scale == 1.0f) && ((__float_as_uint(a.scale) & 0x007fffffu) == 0u);
    auto scaled = [&](float4 q) __attribute__((always_inline)) {   // q / self.scale; a power-of-two scale: exact multiply
      if (pow2) { q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv; }
      else { q.x /= a.scale; q.y /= a.scale; q.z /= a.scale; q.w /= a.scale; }
      return q;
    };
    auto split8 = [&](const float4 u, const float4 v, bf16x8_t* hi, bf16x8_t* lo) __attribute__((always_inline)) {
      bf16x4_t h0, l0, h1, l1;
      split4(u, &h0, &l0);
      split4(v, &h1, &l1);
      *hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      *lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // channel of float4 piece p (0..3) of this lane.  QK == 1 (16x16x4 fp32 tiles, k permuted: the sum over channels
    // does not care which channel meets which MFMA step as long as q and k agree): 16 contiguous channels 16kg + 4p.
    // QK == 2 (16x16x32 bf16 tiles): k-block m = p >> 1 holds channels 32m + 8kg .. +8, pieces 2m, 2m + 1.
    auto chan = [&](int p_) __attribute__((always_inline)) { return QK == 1 ? 16 * kg + 4 * p_ : 32 * (p_ >> 1) + 8 * kg + 4 * (p_ & 1); };
    // Operand path.  A lane's fragment pieces are 16 bytes of ONE row (q row / key l15), neighbouring lanes are rows 9 KB
    // apart: read straight from the token buffer, a wave's dwordx4 load is 64 separate 16-byte requests and the texture
    // addresser works through them one by one.  Ablation builds at B = 256 (406 us): dropping 8 of a wave's 16 key loads
    // -27 us, 6 of its 8 q loads (every wave read all 32 q rows) -23 us; the same scattered loads were what bound the
    // statistics pass of evt_attention_stream.  So both operands now enter COALESCED -- 16 consecutive lanes read one
    // row's 256 bytes -- and are re-read as fragments from LDS: the q rows once per workgroup, each wave's 16 keys of a
    // chunk through a staging block of its own (no workgroup barrier; the next chunk's rows are in flight in registers
    // meanwhile).  The staging blocks live in the A / V tiles, which are idle until the chunk loop.
    constexpr int KP = 68;   // fp32 row pitch: 16 rows x 4 dwords of a fragment read cover the 64 banks once
    constexpr int QPB = 64 + 8;  // bf16 row pitch of the split q planes: 144-byte rows, conflict-free 16-byte reads over 16-lane groups
    constexpr size_t QREG = (QK == 2) ? (size_t)2 * FR * QPB * 2 : (size_t)FR * KP * sizeof(float);   // bytes of the q staging block
    static_assert(QREG % 16 == 0 && QREG + (size_t)4 * 16 * KP * sizeof(float) <= (size_t)(2 * FR + 2 * 64) * P * sizeof(T), "operand staging fits the A / V tiles");
    float* qst = reinterpret_cast<float*>(smem_raw);            // [FR][KP] q rows / scale (exact mode)
    float* kst = reinterpret_cast<float*>(smem_raw + QREG) + wave * 16 * KP;   // [16][KP] this wave's keys of the current chunk
    float4 qg[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 256 * u, row = e >> 4, i = i0 + row;
      qg[u] = *reinterpret_cast<const float4*>(qbase + (uint32_t)((i < a.N ? i : a.N - 1) * rs + 4 * (e & 15)) * 4u);
    }
    // (named native vectors, not a float4 array handed to a lambda: hipcc kept that array in scratch memory)
    typedef float kvec __attribute__((ext_vector_type(4)));
    kvec kr0, kr1, kr2, kr3;       // raw set A: chunks 0 and 2
    kvec ks0, ks1, ks2, ks3;       // raw set B: chunks 1 and 3
    const int krow = lane >> 4, kcol = 4 * (lane & 15);
    const uint32_t kcol_b = (uint32_t)kcol * 4u, rs_b = (uint32_t)rs * 4u;   // byte offsets
    // (rows past N are clamped to the last row -- one v_min each -- so that EVERY chunk's loads are unconditional: the four chunks
    // below are straight-line code and each wait counts exactly the younger requests; inside nested conditions hipcc waited vmcnt(0))
#define EVT_LOAD_KRAW(c0_, r0_, r1_, r2_, r3_)                                                                 \
    do {   /* the wave's 16 key rows of chunk c0_, 4 rows per load */                                          \
      const int j0_ = (c0_) + wave_s * 16 + krow, nl_ = a.N - 1;                                               \
      r0_ = *reinterpret_cast<const kvec*>(kbase_b + ((uint32_t)min(j0_, nl_) * rs_b + kcol_b));               \
      r1_ = *reinterpret_cast<const kvec*>(kbase_b + ((uint32_t)min(j0_ + 4, nl_) * rs_b + kcol_b));           \
      r2_ = *reinterpret_cast<const kvec*>(kbase_b + ((uint32_t)min(j0_ + 8, nl_) * rs_b + kcol_b));           \
      r3_ = *reinterpret_cast<const kvec*>(kbase_b + ((uint32_t)min(j0_ + 12, nl_) * rs_b + kcol_b));          \
    } while (0)
    // TWO raw register sets used alternately (the 32-bit addressing freed the registers: 152 + 16), so that the first TWO key
    // chunks are requested ahead of everything else and every later chunk two chunks ahead: the q.k^T phase is a chain of
    // dependent round trips, and one raw set made it one round trip per chunk.  The score-independent requests of the chunk loop
    // and the epilogue (`prefetch`: 16 two-byte gathers of the old a~ values among them) go out BEHIND the q rows and the first two
    // key chunks: vmcnt retires in order, so in front of them every q.k^T wait was also a wait for the slowest gather.
    EVT_LOAD_KRAW(0, kr0, kr1, kr2, kr3);
    EVT_LOAD_KRAW(QKC, ks0, ks1, ks2, ks3);   // (past N: clamped re-reads of the last row, never used)
    asm volatile("" ::: "memory");
    // Split mode: the q rows are split into bf16 hi | lo ONCE, by the thread that loaded the piece, and staged as two planes (pitch
    // QPB: 144-byte rows, conflict-free 16-byte reads over 16-lane groups) -- every wave reads all 32 rows' fragments, and splitting
    // them after the read was the same 96 VALU instructions four times over, behind the barrier.  Exact mode: fp32 rows as before.
    __bf16* const qsh = reinterpret_cast<__bf16*>(smem_raw);    // [FR][QPB] hi plane (split mode)
    __bf16* const qsl = qsh + FR * QPB;                          // [FR][QPB] lo plane
    float4 qf[2][4];
    bf16x8_t qh[2][2], ql[2][2];
    if (QK == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u;
        bf16x4_t h4, l4;
        split4(scaled(qg[u]), &h4, &l4);
        *reinterpret_cast<bf16x4_t*>(qsh + (e >> 4) * QPB + 4 * (e & 15)) = h4;
        *reinterpret_cast<bf16x4_t*>(qsl + (e >> 4) * QPB + 4 * (e & 15)) = l4;
      }
      __syncthreads();
#pragma unroll
      for (int hr = 0; hr < 2; ++hr)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          qh[hr][m] = *reinterpret_cast<const bf16x8_t*>(qsh + (hr * 16 + l15) * QPB + 32 * m + 8 * kg);
          ql[hr][m] = *reinterpret_cast<const bf16x8_t*>(qsl + (hr * 16 + l15) * QPB + 32 * m + 8 * kg);
        }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u;
        *reinterpret_cast<float4*>(qst + (e >> 4) * KP + 4 * (e & 15)) = scaled(qg[u]);
      }
      __syncthreads();
#pragma unroll
      for (int hr = 0; hr < 2; ++hr)
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) qf[hr][p_] = *reinterpret_cast<const float4*>(qst + (hr * 16 + l15) * KP + chan(p_));
    }
    // raw rows -> the wave's staging block -> this lane's fragment pieces (wave-private: only the wave's own LDS traffic is waited for)
    float4 kf[4];
    float* kst_w = kst + krow * KP + kcol;
    const float* kst_r = kst + l15 * KP;
#define EVT_STAGE_K(r0_, r1_, r2_, r3_)                                                                                          \
    do {                                                                                                       \
      *reinterpret_cast<kvec*>(kst_w) = r0_;                                                                   \
      *reinterpret_cast<kvec*>(kst_w + 4 * KP) = r1_;                                                          \
      *reinterpret_cast<kvec*>(kst_w + 8 * KP) = r2_;                                                          \
      *reinterpret_cast<kvec*>(kst_w + 12 * KP) = r3_;                                                         \
      __builtin_amdgcn_s_waitcnt(0xc07f);                                                                      \
      __builtin_amdgcn_wave_barrier();                                                                         \
      _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) kf[p_] = *reinterpret_cast<const float4*>(kst_r + chan(p_)); \
      __builtin_amdgcn_s_waitcnt(0xc07f);                                                                      \
      __builtin_amdgcn_wave_barrier();                                                                         \
    } while (0)
    auto qk_chunk = [&](int c0, const float4* kf) __attribute__((always_inline)) {
      const int n0 = c0 + wave * 16;
      if (n0 < a.N) {  // wave-uniform
        f32x4_acc sacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (QK == 1) {
#pragma unroll
          for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
            for (int hr = 0; hr < 2; ++hr) {
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].x, qf[hr][p_].x, sacc[hr], 0, 0, 0);
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].y, qf[hr][p_].y, sacc[hr], 0, 0, 0);
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].z, qf[hr][p_].z, sacc[hr], 0, 0, 0);
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].w, qf[hr][p_].w, sacc[hr], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            bf16x8_t kh, kl;
            split8(kf[2 * m], kf[2 * m + 1], &kh, &kl);
#pragma unroll
            for (int hr = 0; hr < 2; ++hr) {
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[hr][m], sacc[hr], 0, 0, 0);
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[hr][m], sacc[hr], 0, 0, 0);
              sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[hr][m], sacc[hr], 0, 0, 0);
            }
          }
        }
        // Keys are the A operand, the q rows the B operand: a lane holds, of query row 16 hr + l15, the scores of the 4 consecutive
        // keys n0 + 4 kg .. + 3 -- one 16-byte LDS store per row group instead of four 4-byte ones per row.  (Columns in [N, EP) take
        // the scores of the clamped / staged rows past N: nobody reads them.)
        const int j4 = n0 + 4 * kg;
        if (j4 < EP) {
#pragma unroll
          for (int hr = 0; hr < 2; ++hr)
            *reinterpret_cast<f32x4_acc*>(et + (hr * 16 + l15) * EP + j4) = sacc[hr];
        }
      }
    };
    // N <= 256 (entry-point check): at most four chunks, written out as straight-line code -- qk_chunk skips a chunk past N
    // (wave-uniform), its (clamped) loads and staging are harmless.  Issue order (vmcnt retires in order): chunks 0, 1 | chunk 2
    // behind the staging of 0, chunk 3 behind the staging of 1 | then the score-independent requests of the chunk loop and the
    // epilogue (`prefetch`), so that no q.k^T wait includes them; they land during the last two chunks and the statistics.
    static_assert(QKC == 64, "four chunks of 64 keys cover N <= 256");
    EVT_STAGE_K(kr0, kr1, kr2, kr3);
    EVT_LOAD_KRAW(2 * QKC, kr0, kr1, kr2, kr3);   // set A is free again
    qk_chunk(0, kf);
    EVT_STAGE_K(ks0, ks1, ks2, ks3);
    EVT_LOAD_KRAW(3 * QKC, ks0, ks1, ks2, ks3);
    asm volatile("" ::: "memory");
    prefetch();
    asm volatile("" ::: "memory");
    qk_chunk(QKC, kf);
    EVT_STAGE_K(kr0, kr1, kr2, kr3);
    qk_chunk(2 * QKC, kf);
    EVT_STAGE_K(ks0, ks1, ks2, ks3);
    qk_chunk(3 * QKC, kf);
#undef EVT_LOAD_KRAW
#undef EVT_STAGE_K
    __syncthreads();
    ATT_TICK(0);   // prefetch issue + q.k^T
  }
  if (NREG > 0) {
    float xv[8][NREG > 0 ? NREG : 1];
    // (wave-uniform) only a row's LAST register can reach past Nk: the others need no clamp, no mask and no store predicate, and
    // their LDS addresses are one per-row base + immediate offsets
    const bool fullregs = QK && a.Nk > 64 * (NREG - 1);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave_s * 8 + rr, i = i0 + r;
      const float* prow = QK ? (et + r * EP) : (prod + (int64_t)(i < a.N ? i : 0) * a.Nk);
#pragma unroll
      for (int u = 0; u < NREG; ++u) {
        // branch-free: a clamped address keeps all 8*NREG loads of the wave in flight together (a predicated load
        // inside `if (j < Nk)` compiles to load + s_waitcnt vmcnt(0) per element)
        const int j = lane + 64 * u;
        if (fullregs && u < NREG - 1) {
          xv[rr][u] = prow[j];
        } else {
          const float x = prow[j < a.Nk ? j : a.Nk - 1];
          xv[rr][u] = (j < a.Nk) ? x : -INFINITY;
        }
      }
    }
    if (QK) {   // every lane holds its scores before the row's exps overwrite them in the same tile
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    if (!QK) prefetch();
    asm volatile("" ::: "memory");   // scheduling fence: the requests above are issued before the statistics below
    if (rel) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const float* rv = relv + (wave * 8 + rr) * nrel;
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
          const int j = lane + 64 * u;
          if (j < a.Nk) { const int ky = fast_div(j, inv_gw); xv[rr][u] = (xv[rr][u] + rv[ky]) + rv[a.gh + j - ky * a.gw]; }
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float mx = xv[rr][0];
#pragma unroll
      for (int u = 1; u < NREG; ++u) mx = fmaxf(mx, xv[rr][u]);
      rmax[rr] = mx;
    }
    wave_max_dpp_rows<8>(rmax);   // the eight rows' reductions step by step (same bits as wave_max_dpp per row)
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < NREG; ++u) {
        const float e = fast_exp(xv[rr][u] - rmax[rr]);  // exp(-inf) == 0 past N
        if ((fullregs && u < NREG - 1) || lane + 64 * u < a.Nk) et[(wave_s * 8 + rr) * EP + lane + 64 * u] = e;
        sum += e;
      }
      rsum[rr] = sum;
    }
    wave_sum_dpp_rows<8>(rsum);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();  // each wave only ever reads back its own 8 rows
  } else {
    ATT_TICK(9);   // (streamed path) everything before the row pass: index loads, rel-pos terms
    // any N: ONE pass over the row with an online (running max / rescaled sum) softmax, 8 independent
    // 256-byte wave loads in flight per step so the cold HBM stream is not latency-serialised.
    if (vvec) load_v(0);
    load_pv();
    if ((a.Nk & 3) == 0) {
      // Two rows at a time, each in its own register buffer: the loads of the next 2048-column step are requested while
      // the current one is reduced (in-kernel phase timing, ViTDet 42 x 42: the row-after-row version spent 19k ticks per row,
      // most of them waiting for the 8 KB it had just asked for).  Every request is unconditional (rows past N and the
      // step after the last one re-read a valid address), so the waits can count on the younger requests being in flight.
      const int nsteps = (a.Nk + 2047) / 2048;
      auto issue = [&](int rr, int js, float4* x4) __attribute__((always_inline)) {
        const int i = i0 + wave * 8 + rr;
        const float* prow = prod + (int64_t)(i < a.N ? i : a.N - 1) * a.Nk;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = js * 2048 + (lane + 64 * u) * 4;
          x4[u] = *reinterpret_cast<const float4*>(prow + (j < a.Nk ? j : 0));
        }
      };
      auto consume = [&](int rr, int js, float4* x4, float& mx, float& sum) __attribute__((always_inline)) {
        const float* rv = relv + (wave * 8 + rr) * nrel;
        const int j0 = js * 2048;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + (lane + 64 * u) * 4;
          if (j >= a.Nk) x4[u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
          else if (rel) {
            float* xe = reinterpret_cast<float*>(&x4[u]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int ky = fast_div(j + q, inv_gw); xe[q] = (xe[q] + rv[ky]) + rv[a.gh + j + q - ky * a.gw]; }
          }
        }
        float cm = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) cm = fmaxf(cm, fmaxf(fmaxf(x4[u].x, x4[u].y), fmaxf(x4[u].z, x4[u].w)));
        const float nm = fmaxf(mx, cm);
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          part += (fast_exp(x4[u].x - nm) + fast_exp(x4[u].y - nm)) + (fast_exp(x4[u].z - nm) + fast_exp(x4[u].w - nm));
        sum = (nm == -INFINITY) ? 0.f : sum * fast_exp(mx - nm) + part;
        mx = nm;
      };
      auto finish_row = [&](int rr, float mx, float sum) __attribute__((always_inline)) {
        const bool live = i0 + wave * 8 + rr < a.N;   // wave-uniform
        const float wmx = wave_max_dpp(mx);
        sum = (mx == -INFINITY) ? 0.f : sum * fast_exp(mx - wmx);
        const float ws = wave_sum_dpp(sum);
        rmax[rr] = live ? wmx : 0.f;
        rsum[rr] = live ? ws : 1.f;
      };
      float4 xa[8], xb[8];
      issue(0, 0, xa);
#pragma unroll
      for (int rp = 0; rp < 8; rp += 2) {
        float mxa = -INFINITY, suma = 0.f, mxb = -INFINITY, sumb = 0.f;
        for (int js = 0; js < nsteps; ++js) {
          issue(rp + 1, js, xb);
          consume(rp, js, xa, mxa, suma);
          const bool more = js + 1 < nsteps;
          issue(more ? rp : (rp + 2 < 8 ? rp + 2 : rp), more ? js + 1 : 0, xa);   // (the very last one is a dummy re-read)
          consume(rp + 1, js, xb, mxb, sumb);
        }
        finish_row(rp, mxa, suma);
        finish_row(rp + 1, mxb, sumb);
      }
    } else
#pragma unroll 1
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave * 8 + rr, i = i0 + r;
      rmax[rr] = 0.f; rsum[rr] = 1.f;
      if (i >= a.N) continue;  // wave-uniform
      const float* rv = relv + r * nrel;
      const float* prow = prod + (int64_t)i * a.Nk;
      float mx = -INFINITY, sum = 0.f;
      for (int j0 = 0; j0 < a.Nk; j0 += 512) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + lane + 64 * u;
          x[u] = (j < a.Nk) ? prow[j] : -INFINITY;
        }
        if (rel) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + lane + 64 * u;
            if (j < a.Nk) { const int ky = fast_div(j, inv_gw); x[u] = (x[u] + rv[ky]) + rv[a.gh + j - ky * a.gw]; }
          }
        }
        float cm = x[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) cm = fmaxf(cm, x[u]);
        const float nm = fmaxf(mx, cm);
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) part += fast_exp(x[u] - nm);   // exp(-inf) == 0 for masked / empty
        sum = (nm == -INFINITY) ? 0.f : sum * fast_exp(mx - nm) + part;
        mx = nm;
      }
      // combine the 64 per-lane (max, sum) pairs
      const float wmx = wave_max_dpp(mx);
      sum = (mx == -INFINITY) ? 0.f : sum * fast_exp(mx - wmx);
      rmax[rr] = wmx;
      rsum[rr] = wave_sum_dpp(sum);
    }
  }

  ATT_TICK(1);   // softmax statistics
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int prodsel = wave >> 1;  // 0: a~ . dv~   1: da~ . v_old
  const int half = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  static_assert(FKC == 64, "one selected column per lane and chunk");
  // softmax normaliser as a reciprocal, one IEEE division per row instead of one per gathered element: e * (1/sum)
  // is within 1 ulp of e / sum before the rounding to the store type
  float rinv[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) rinv[rr] = 1.0f / rsum[rr];
  // One chunk of 64 selected columns.  jcol: this lane's column (-1 past count); oldp: the 8 old a~ values of the
  // lane's column if they were requested ahead, else nullptr (loaded here, all 8 in flight, clamped addresses).
  auto do_chunk = [&](int k0, int jcol, const float* oldp, const bool set1, const bool ahead) __attribute__((always_inline)) {
    // set1: this chunk's V pieces are in vpd1 / vpo1 (requested two chunks ahead); ahead: the NEXT chunk's pieces have
    // already been requested (into the other set), nothing to request here
    // ---- phase 2a: gather the chunk's columns for this wave's 8 rows (A delta gate) -------------
    // `full`: predicated stores sit in exec-masked branches; hipcc cannot count them and the next counted wait -- the epilogue's,
    // for the A.v state rows requested long ago -- became vmcnt(0): a whole store round trip in the open.
    const bool full = i0 + FR <= a.N && k0 + FKC <= cnt;
    const int js = jcol >= 0 ? jcol : 0;
    float oldv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int i = i0 + wave_s * 8 + rr;
      oldv[rr] = oldp ? oldp[rr] : st_load(st_off(i < a.N ? i : 0, js));
    }
    // the 8 exp-tile reads of the lane's column go out together and are waited for once (hipcc otherwise sinks each read
    // into the predicated store block of its row: eight serialised LDS round trips per chunk)
    float ev[8];
    if (NREG > 0) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) ev[rr] = et[(wave_s * 8 + rr) * EP + js];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) asm volatile("" : "+v"(ev[rr]));
    } else {   // streamed path: the 8 score gathers likewise (eight serialised HBM round trips per chunk otherwise)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int i = i0 + wave * 8 + rr;
        ev[rr] = prod[(int64_t)(i < a.N ? i : 0) * a.Nk + js];
      }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) asm volatile("" : "+v"(ev[rr]));
    }
    if (NREG > 0 && full) {
      // every row of the tile and every column of the chunk exists: straight-line code for the eight rows (no per-row exec-mask
      // block), the rounded values' bits reused for the stores
      const uint32_t jb = (uint32_t)jcol * (uint32_t)sizeof(T);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = wave_s * 8 + rr;
        const float an = Store<T>::round(ev[rr] * rinv[rr]);
        const float ad = Store<T>::round(an - oldv[rr]);
        st_store((uint32_t)((i0 + r) * a.Nk) * (uint32_t)sizeof(T) + jb, an);
        Store<T>::store(An + r * P + lane, an);
        Store<T>::store(Ad + r * P + lane, ad);
      }
    } else {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave_s * 8 + rr, i = i0 + r;
      const bool ok = i < a.N && jcol >= 0;
      float e;
      if (NREG > 0) {
        e = ev[rr];
      } else {
        float x = ev[rr];
        if (rel) { const float* rv = relv + r * nrel; const int ky = fast_div(js, inv_gw); x = (x + rv[ky]) + rv[a.gh + js - ky * a.gw]; }
        e = fast_exp(x - rmax[rr]);
      }
      float an = Store<T>::round(e * rinv[rr]);
      float ad = Store<T>::round(an - oldv[rr]);
      if (full) {   // (wave-uniform) every row of the tile and every column of the chunk exists: the reference store is unconditional
        st_store(st_off(i, jcol), an);
      } else {
        if (ok) st_store(st_off(i, jcol), an);
        an = ok ? an : 0.f;
        ad = ok ? ad : 0.f;
      }
      Store<T>::store(An + r * P + lane, an);
      Store<T>::store(Ad + r * P + lane, ad);
    }
    }
    ATT_TICK(5);   // 2a: old values, exp tile reads, A gate, scattered state stores, LDS stores
    // ---- phase 2b: stage the chunk of dv~^T and v_old^T (k contiguous; requested one chunk ahead) ----
    if (vvec) {
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC;
        *reinterpret_cast<uint4*>(Vd + d * P + jj) = set1 ? vpd1[it] : vpd[it];
        *reinterpret_cast<uint4*>(Vo + d * P + jj) = set1 ? vpo1[it] : vpo[it];
      }
      if (!ahead && k0 + FKC < cnt) load_v(k0 + FKC);   // flies during the MFMA sweep of this chunk
    } else {
      for (int e = tid; e < DHC * FKC; e += 256) {
        const int d = e / FKC, jj = e - d * FKC, kk = k0 + jj;
        float vd = 0.f, vo = 0.f;
        if (kk < cnt) {
          vd = Store<T>::load(Vg_d + (int64_t)d * a.kcap + kk);
          vo = Store<T>::load(Vg_o + (int64_t)d * a.kcap + kk);
        }
        Store<T>::store(Vd + d * P + jj, vd);
        Store<T>::store(Vo + d * P + jj, vo);
      }
    }
    ATT_TICK(6);   // 2b: V staging, next V request
    __syncthreads();
    ATT_TICK(2);   // barrier after staging
    // ---- phase 3: matrix cores ------------------------------------------------------------------
    const T* At = (prodsel == 0 ? An : Ad) + lr * P;
    const T* Vt = (prodsel == 0 ? Vd : Vo);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      acc[t] = Tile<T>::sweep(At, Vt + ((half + 2 * t) * 32 + lr) * P, lh, acc[t]);
    __syncthreads();
    ATT_TICK(3);   // MFMA sweep (+ barrier)
  };
  if (PF > 0) {
    // The A.v state rows and the next gate's reference (requested with `prefetch`, used in the epilogue) are claimed HERE, while no
    // store is in flight: the reference stores of the chunk loop sit in conditional blocks hipcc cannot count, so the first wait
    // behind them is vmcnt(0) -- in the epilogue that was a whole store round trip in the open.
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
#pragma unroll
      for (int q = 0; q < (int)((8 * sizeof(T)) / 16); ++q)
        asm volatile("" : "+v"(pvr[it].u[q].x), "+v"(pvr[it].u[q].y), "+v"(pvr[it].u[q].z), "+v"(pvr[it].u[q].w));
      if (a.norm_ref != nullptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          asm volatile("" : "+v"(nrr[it].v[q].x), "+v"(nrr[it].v[q].y), "+v"(nrr[it].v[q].z), "+v"(nrr[it].v[q].w));
      }
    }
    // the second chunk's V pieces go out now (the q.k^T fragment registers are free again) instead of during the first
    // chunk's MFMA sweep: at r = 128 (two chunks) nothing inside the chunk loop waits for HBM / L2 any more
    const bool two = vvec && cnt > FKC;
    if (two) load_v_into(FKC, vpd1, vpo1);
    if (cnt > 0) do_chunk(0, jpf[0], oldpf[0], false, two);
    if (cnt > FKC) {
      if (two) do_chunk(FKC, jpf[PF > 1 ? 1 : 0], oldpf[PF > 1 ? 1 : 0], true, false);
      else do_chunk(FKC, jpf[PF > 1 ? 1 : 0], oldpf[PF > 1 ? 1 : 0], false, false);
    }
  }
  for (int k0 = PF * FKC; k0 < cnt; k0 += FKC) {
    const int kk = k0 + lane;
    const int j = ix[kk < cnt ? kk : 0];
    do_chunk(k0, kk < cnt ? j : -1, nullptr, false, false);
  }

  // ---- phase 4: state += round(acc1); state += round(acc2); heads merged on write ---------------
  // Both rounded products go through LDS so that the state read-modify-write and the fp32 output are
  // whole-row 16-byte accesses by all 256 threads (8 consecutive channels per thread).
  {
    float* redp = (prodsel == 0) ? red1 : red2;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        redp[row * RDP + (half + 2 * t) * 32 + lr] = Store<T>::round(acc[t][r]);
      }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int e = tid + 256 * it, row = e / (DHC / 8), c8 = (e - row * (DHC / 8)) * 8;
    const int i = i0 + row;
    const bool ok = i < a.N;
    const uint32_t o = (uint32_t)((ok ? i : a.N - 1) * a.D + c8);   // element offset from this clip's rows / this head's channels
    Pv8 st8 = pvr[it];
    union { float4 v[2]; float f[8]; } o8, r1, r2;
    // the thread's 8 channels of both rounded products as 16-byte LDS reads (rows are DHC floats apart: the same banks -- element by
    // element the 8 rows of a wave collided 8-way on every read)
    r1.v[0] = *reinterpret_cast<const float4*>(red1 + row * RDP + c8);
    r1.v[1] = *reinterpret_cast<const float4*>(red1 + row * RDP + c8 + 4);
    r2.v[0] = *reinterpret_cast<const float4*>(red2 + row * RDP + c8);
    r2.v[1] = *reinterpret_cast<const float4*>(red2 + row * RDP + c8 + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = Store<T>::round(Store<T>::load(&st8.t[q]) + r1.f[q]);  // += a~ . dv~
      v = Store<T>::round(v + r2.f[q]);                                 // += da~ . v_old
      Store<T>::store(&st8.t[q], v);
      o8.f[q] = v;
    }
    if (ok) {
#pragma unroll
      for (int q = 0; q < (int)((8 * sizeof(T)) / 16); ++q)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(pv_bh) + (o * (uint32_t)sizeof(T) + 16u * q)) = st8.u[q];
      if (a.out_f32 != nullptr) {   // wave-uniform; NULL: the caller reads the (identical) values from the A.v state
        char* const ob = reinterpret_cast<char*>(a.out_f32 + ((int64_t)b * a.N * a.D + h * DHC));
        *reinterpret_cast<float4*>(ob + o * 4u) = o8.v[0];
        *reinterpret_cast<float4*>(ob + (o * 4u + 16u)) = o8.v[1];
      }
    }
    if (a.norm_parts != nullptr) {   // wave-uniform
      // ||out - ref||^2 over this head's channels: the dh/8 threads of a row are consecutive lanes; their sums are
      // combined in a fixed butterfly order (deterministic), the first lane of the group writes the head's partial
      float ss = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = o8.f[q] - nrr[it].f[q]; ss = fmaf(d, d, ss); }
#pragma unroll
      for (int m = 1; m < DHC / 8; m <<= 1) ss += __shfl_xor(ss, m, 64);
      if (ok && (tid & (DHC / 8 - 1)) == 0) (a.norm_parts + ((int64_t)b * a.N * a.H + h))[(uint32_t)(i * a.H)] = ss;
    }
  }
#ifdef EVT_PROF
  ATT_TICK(4);   // epilogue
  if (prof_on && lane == 0)
    for (int q = 0; q < 16; ++q) evt_prof_attn_buf[q] = prof_acc[q];
#endif
}

template <typename T, int TPW, int NREG, int QK = 0>
void launch_fused_inst(const FusedArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  EVT_ALLOW_LDS((softmax_av_gated_kernel<T, TPW, NREG, QK>), lds);
  hipLaunchKernelGGL((softmax_av_gated_kernel<T, TPW, NREG, QK>), grid, dim3(256), lds, s, a);
}

template <typename T>
int launch_fused(const FusedArgs& a, void* stream) {
  constexpr int P = Tile<T>::PITCH;
  const int nreg = (a.Nk + 63) / 64;
  const size_t tile_e = nreg <= 4 ? (size_t)FR * fused_ep(a.Nk) : 0;   // exp tile; q rows and the rounded-product
  static_assert(2 * FR * sizeof(float) <= 2 * P * sizeof(T), "red tiles must fit in the V tiles");  // tiles alias Vd/Vo
  const size_t lds = (size_t)(2 * FR + 2 * a.dh) * P * sizeof(T) + (tile_e + FR * (a.gh + a.gw)) * sizeof(float);
  const dim3 grid((a.N + FR - 1) / FR, a.B * a.H);
  if (grid.y == 0) return EVT_OK;
  hipStream_t s = evt_stream(stream);
  if (a.product == nullptr) {   // QK mode (validated by the entry point: dh == 64, N == Nk <= 256, kcap > 0)
    if (a.qk_split) launch_fused_inst<T, 1, 4, 2>(a, grid, lds, s);
    else launch_fused_inst<T, 1, 4, 1>(a, grid, lds, s);
  } else if (a.dh == 64) {
    if (nreg <= 4 && a.kcap > 0) launch_fused_inst<T, 1, 4>(a, grid, lds, s);
    else launch_fused_inst<T, 1, 0>(a, grid, lds, s);
  } else {
    if (nreg <= 4 && a.kcap > 0) launch_fused_inst<T, 2, 4>(a, grid, lds, s);
    else launch_fused_inst<T, 2, 0>(a, grid, lds, s);
  }
  return evt_check_launch("evt_softmax_av_gated");
}

}  // namespace

extern "C" int evt_softmax_av_gated(const evt_softmax_av_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_softmax_av_gated: null descriptor");
  EVT_REQUIRE(d->a_state && d->idx && d->v_delta_t && d->v_old_t && d->pv, EVT_ERR_BAD_ARG, "evt_softmax_av_gated: null pointer");
  EVT_REQUIRE(d->out_f32 != nullptr || d->store != EVT_F32, EVT_ERR_BAD_ARG,
              "evt_softmax_av_gated: out_f32 may only be omitted with a 16-bit store type (the output then IS the pv state)");
  if (d->product == nullptr) {  // QK mode: scores from the token buffer
    EVT_REQUIRE(d->qkv != nullptr && d->scale > 0.f, EVT_ERR_BAD_ARG, "evt_softmax_av_gated: product == NULL needs qkv and scale");
    EVT_REQUIRE(d->dh == 64 && d->Nk == d->N && d->N <= 256 && d->kcap > 0, EVT_ERR_BAD_SHAPE,
                "evt_softmax_av_gated: in-kernel q.k^T needs head dim 64, N == Nk <= 256, kcap > 0 (dh=%d N=%d Nk=%d kcap=%d)",
                d->dh, d->N, d->Nk, d->kcap);
  }
  EVT_REQUIRE(d->B >= 0 && d->H > 0 && d->N > 0 && d->kcap >= 0 && d->D == d->H * d->dh, EVT_ERR_BAD_ARG,
              "evt_softmax_av_gated: bad sizes");
  EVT_REQUIRE(d->dh == 64 || d->dh == 128, EVT_ERR_BAD_SHAPE,
              "evt_softmax_av_gated: head dim %d not supported by the fused kernel (64 or 128); use evt_softmax_gate + evt_av",
              d->dh);
  EVT_REQUIRE((d->rel_y == nullptr) == (d->rel_x == nullptr), EVT_ERR_BAD_ARG, "evt_softmax_av_gated: rel_y/rel_x");
  EVT_REQUIRE(d->Nk > 0, EVT_ERR_BAD_ARG, "evt_softmax_av_gated: Nk=%d", d->Nk);
  if (d->rel_y) {
    EVT_REQUIRE(d->qkv != nullptr && d->gh > 0 && d->gw > 0 && d->gh * d->gw == d->Nk && d->qw > 0 && d->N % d->qw == 0,
                EVT_ERR_BAD_SHAPE, "evt_softmax_av_gated: rel-pos key grid %dx%d / query width %d do not match N=%d Nk=%d",
                d->gh, d->gw, d->qw, d->N, d->Nk);
  }
  FusedArgs a{d->product, d->qkv, d->rel_y, d->rel_x, d->a_state, d->idx, d->count, d->v_delta_t, d->v_old_t,
              d->pv, d->out_f32, d->B, d->H, d->N, d->Nk, d->D, d->dh, d->kcap, d->rel_y ? d->gh : 0, d->rel_y ? d->gw : 0,
              d->rel_y ? d->qw : 1, d->scale, d->qk_split, d->norm_ref, d->norm_parts, d->rel_y ? d->rel_terms : nullptr};
  EVT_REQUIRE((d->norm_ref == nullptr) == (d->norm_parts == nullptr), EVT_ERR_BAD_ARG, "evt_softmax_av_gated: norm_ref / norm_parts come together");
  // the kernel addresses a head's N x Nk reference, a clip's (N, D) states and its (D, kcap) value operands by 32-bit byte offsets
  EVT_REQUIRE((int64_t)d->N * d->Nk * 4 < (1ll << 32) && (int64_t)d->N * d->D * 4 < (1ll << 32) && (int64_t)d->D * d->kcap * 4 < (1ll << 32) &&
              (int64_t)d->N * 3 * d->D * 4 < (1ll << 32), EVT_ERR_BAD_SHAPE,
              "evt_softmax_av_gated: a head's reference (N x Nk), a clip's state (N x D) or its token buffer exceeds 4 GB (N=%d Nk=%d D=%d kcap=%d)",
              d->N, d->Nk, d->D, d->kcap);
  EVT_DISPATCH_STORE(d->store, T, { return launch_fused<T>(a, stream); });
  return EVT_OK;
}

namespace {

// evt_rel_terms: one workgroup per (clip*head, query-grid row y or column x).  blockIdx.x < qh: the qw queries of row y
// against rel_y[y] (gh x 64); else the qh queries of column x against rel_x[x] (gw x 64).  Queries and table slice are
// staged in LDS (row pitch 68 floats); every thread accumulates its (query, key) dots in the order of the in-kernel
// version (two interleaved partial sums over 16-byte pieces).
__global__ __launch_bounds__(256) void rel_terms_kernel(const float* __restrict__ qkv, const float* __restrict__ rel_y,
                                                        const float* __restrict__ rel_x, int H, int N, int D, int gh, int gw,
                                                        int qw, float* __restrict__ terms) {
  constexpr int DH = 64, LP = DH + 4;
  extern __shared__ __attribute__((aligned(16))) float rt_smem[];
  const int qh = N / qw, nrel = gh + gw;
  const bool is_y = (int)blockIdx.x < qh;
  const int pos = is_y ? (int)blockIdx.x : (int)blockIdx.x - qh;   // y, or x
  const int nq = is_y ? qw : qh, nkey = is_y ? gh : gw;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  float* qs = rt_smem;               // [nq][LP]
  float* ts = rt_smem + nq * LP;     // [nkey][LP]
  const float* tab = is_y ? rel_y + (int64_t)pos * gh * DH : rel_x + (int64_t)pos * gw * DH;
  const int tid = threadIdx.x;
  for (int e = tid; e < nq * (DH / 4); e += 256) {
    const int r = e / (DH / 4), c4 = e - r * (DH / 4);
    const int i = is_y ? pos * qw + r : r * qw + pos;
    *reinterpret_cast<float4*>(qs + r * LP + c4 * 4) =
        *reinterpret_cast<const float4*>(qkv + ((int64_t)b * N + i) * 3 * D + h * DH + c4 * 4);
  }
  for (int e = tid; e < nkey * (DH / 4); e += 256) {
    const int r = e / (DH / 4), c4 = e - r * (DH / 4);
    *reinterpret_cast<float4*>(ts + r * LP + c4 * 4) = *reinterpret_cast<const float4*>(tab + (int64_t)r * DH + c4 * 4);
  }
  __syncthreads();
  // 4 x 4 outputs per thread: 8 LDS reads of 16 bytes per 4 channels instead of 32
  const int tk = (nkey + 3) / 4, tiles = ((nq + 3) / 4) * tk;
  for (int e = tid; e < tiles; e += 256) {
    const int r0 = (e / tk) * 4, k0 = (e - (e / tk) * tk) * 4;
    float s0[4][4], s1[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) s0[x][y] = s1[x][y] = 0.f;
#pragma unroll 2
    for (int d = 0; d < DH / 4; d += 2) {
      float4 qa[4], qb[4], ta[4], tb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float* q = qs + min(r0 + x, nq - 1) * LP + 4 * d;
        const float* t = ts + min(k0 + x, nkey - 1) * LP + 4 * d;
        qa[x] = *reinterpret_cast<const float4*>(q);
        qb[x] = *reinterpret_cast<const float4*>(q + 4);
        ta[x] = *reinterpret_cast<const float4*>(t);
        tb[x] = *reinterpret_cast<const float4*>(t + 4);
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          s0[x][y] += qa[x].x * ta[y].x + qa[x].y * ta[y].y + qa[x].z * ta[y].z + qa[x].w * ta[y].w;
          s1[x][y] += qb[x].x * tb[y].x + qb[x].y * tb[y].y + qb[x].z * tb[y].z + qb[x].w * tb[y].w;
        }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int r = r0 + x;
      if (r >= nq) continue;
      const int i = is_y ? pos * qw + r : r * qw + pos;
      float* dst = terms + ((int64_t)bh * N + i) * nrel + (is_y ? 0 : gh);
#pragma unroll
      for (int y = 0; y < 4; ++y)
        if (k0 + y < nkey) dst[k0 + y] = s0[x][y] + s1[x][y];
    }
  }
}

// The same terms on the matrix cores in split precision (q and the table rows as bf16 hi + lo, three
// v_mfma_f32_16x16x32_bf16 per product: the arithmetic of the scores they are added to).  The VALU version keeps 121 of a
// workgroup's 256 threads busy with 1024 dependent FMAs each at 42 x 42 (14 us per launch at 672^2, 30 us at 1024^2: four
// launches per frame of one stream).  Wave w owns the 16-query tiles w, w + 4, ...; a lane's accumulator holds queries
// 4 kg + r of the tile against key l15, so the 16 lanes of a group store 16 consecutive keys of one query row.
__global__ __launch_bounds__(256) void rel_terms_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ rel_y,
                                                             const float* __restrict__ rel_x, int H, int N, int D, int gh, int gw,
                                                             int qw, float* __restrict__ terms) {
  extern __shared__ __attribute__((aligned(16))) float rt_smem[];
  evt_rel_terms_mfma_role(qkv, rel_y, rel_x, H, N, D, gh, gw, qw, terms, (int)blockIdx.x, (int)blockIdx.y, rt_smem);   // evt_prep_roles.h
}

}  // namespace

extern "C" int evt_rel_terms(const float* qkv, const float* rel_y, const float* rel_x, int32_t B, int32_t H, int32_t N,
                             int32_t D, int32_t gh, int32_t gw, int32_t qw, int32_t split, float* terms, void* stream) {
  EVT_REQUIRE(qkv && rel_y && rel_x && terms, EVT_ERR_BAD_ARG, "evt_rel_terms: null pointer");
  EVT_REQUIRE(B >= 0 && H > 0 && N > 0 && gh > 0 && gw > 0 && qw > 0 && N % qw == 0 && D == H * 64, EVT_ERR_BAD_SHAPE,
              "evt_rel_terms: head dim 64 and N = qh * qw required (B=%d H=%d N=%d D=%d qw=%d)", B, H, N, D, qw);
  const int qh = N / qw;
  const size_t lds = (size_t)(std::max(qw, qh) + std::max(gh, gw)) * 68 * sizeof(float);
  EVT_REQUIRE(lds <= 160 * 1024, EVT_ERR_BAD_SHAPE, "evt_rel_terms: grid %dx%d too large", qh, qw);
  if (B == 0) return EVT_OK;
  if (split) {
    EVT_ALLOW_LDS(rel_terms_mfma_kernel, lds);
    hipLaunchKernelGGL(rel_terms_mfma_kernel, dim3(qh + qw, B * H), dim3(256), lds, evt_stream(stream), qkv, rel_y, rel_x, H, N, D,
                       gh, gw, qw, terms);
  } else {
    EVT_ALLOW_LDS(rel_terms_kernel, lds);
    hipLaunchKernelGGL(rel_terms_kernel, dim3(qh + qw, B * H), dim3(256), lds, evt_stream(stream), qkv, rel_y, rel_x, H, N, D, gh, gw,
                       qw, terms);
  }
  return evt_check_launch("evt_rel_terms");
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof_attn(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_attn_buf), sizeof(unsigned long long) * 16);
}
#endif
