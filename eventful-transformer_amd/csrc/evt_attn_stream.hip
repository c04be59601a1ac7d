// evt_attn_stream.hip -- attention of EventfulBlock for LARGE token counts (N > 256: the global blocks of ViTDet,
// N = 1764 / 4096), scores computed in the kernel.  One launch per frame and block:
//
//   gated frame   softmax statistics of all N columns + [rel-pos]                    blocks.py:518-522
//                 -> the k selected columns: a~, da~ = a~ - p[:, idx], p[:, idx] = a~  modules.py:187-201 ("col")
//                 -> state += round(a~ . dv~); state += round(da~ . v_old), heads merged modules.py:285-295
//   first frame   a = round(softmax(..)) -> matmul_gate reference p; state = out = round(a . round(v))
//                                                                                     modules.py:183-185, 277-283
//
// Why a second fused kernel.  For these shapes evt_softmax_av_gated read the (B,H,N,N) fp32 q.k^T state once per frame
// (149 MB per block at 672^2, 805 MB at 1024^2) after evt_qk had patched its k rows and k columns (86 MB of scattered
// writes at 672^2), and gathered / scattered the gate reference p with one 4-byte (2-byte) access per element at a
// stride of a whole state row: one video stream spent 193 us (672^2) / 824 us (1024^2) per global block in evt_qk +
// evt_softmax_av_gated.  Here
//   * the scores are never stored: a workgroup (32 or 48 query rows) recomputes (q / scale) k^T for its rows against all N keys on
//     the matrix cores, straight from the token buffer (q and k go from L2 into MFMA fragments, bf16 hi/lo split
//     products like evt_qk, or exact fp32 products with qk_split = 0), keeping only running (max, sum) pairs -- 2 N^2 D
//     FLOP per block (14 GFLOP of bf16 MFMA work at 672^2: microseconds) instead of the state traffic, and the q.k^T
//     state, evt_qk and the state's memory (4 x 149 MB / 4 x 805 MB per stream) disappear;
//   * the scores of the k selected columns are then computed once more (k x 64 key rows gathered through idx);
//   * the gate reference is kept TRANSPOSED, (B,H,Nk,N): the rows of a workgroup are contiguous bytes of a selected
//     column (16 lanes read / rewrite 16 consecutive rows of one column).
// Score layout.  The key rows are the A operand and the query rows the B operand of v_mfma_f32_16x16x32_bf16, so a lane
// holds, of ONE query row, the scores of 4 consecutive keys per tile: the online softmax rescales once per 4 scores (one
// exponential per score + one per group; a lane owning 8 different rows paid a rescale per score), keeps (max, sum) of
// 2-4 rows instead of 8, and the cross-lane combine is two shuffle steps.  In-kernel phase timing (scripts/
// onestream_bench.py, -DEVT_PROF) of the row-per-lane-group version: 49 % of a workgroup's life in the statistics pass at
// 1700 cycles per 64-key chunk and wave.
// Tile height.  672 workgroups of 32 rows on 512 slots (2 per CU) run in 1.31 rounds, i.e. the time of two: the launcher
// picks 32 or 48 rows (NHR = 2, 3 MFMA row groups of 16) minimising rounds x per-workgroup cost; 1764 tokens x 12
// heads: 48 rows = 444 workgroups = one round.
// The A.v part (LDS tiles of a~ / da~, both products on the matrix cores, state read-modify-write epilogue, per-head
// partial ||out - ref||^2 for the projection gate) is that of evt_softmax_av_gated; every rounding point of the reference
// is kept.  Head dim 64, un-pooled keys.
#include "evt_attn_tiles.h"
#include "evt_prep_roles.h"
#include <algorithm>
#include <stdlib.h>

namespace {

struct StreamArgs {
  const float* qkv; const float* rel_terms;
  void* a_state_t; const int32_t* idx; const int32_t* count;
  const void* v_delta_t; const void* v_old_t; const void* v_state; void* pv; float* out_f32;
  const float* norm_ref; float* norm_parts;
  void* k_split;
  int B, H, N, D, kcap, gh, gw;
  float scale;
  int k_split_ready;   // the key plane has been written by evt_stream_prep: no pre-kernel
  // Keys (and values): Nk rows of `ksrc`, row stride k_rs floats, key channels at k_off.  Un-pooled: the packed token buffer
  // (Nk = N, k_rs = 3 D, k_off = D); pooled (blocks.py:303-326): the (B,Nk,2D) buffer of evt_pool_kv (k_rs = 2 D, k_off = 0).
  const float* ksrc; int Nk, k_rs, k_off;
};

typedef float f32x4_acc __attribute__((ext_vector_type(4)));

#ifdef EVT_PROF   // phase timing of wave 0 of one workgroup (scripts/onestream_bench.py with a -DEVT_PROF build)
__device__ unsigned long long evt_prof_stream_buf[16];
#define STR_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define STR_TICK(slot) do { } while (0)
#endif

// four consecutive elements of the store type as ONE LDS access (4 selected columns of a row of the a~ / da~ tiles)
template <typename T> struct Quad { union { uint2 v; T t[4]; }; };
template <> struct Quad<float> { union { float4 v; float t[4]; }; };

constexpr int SDH = 64;   // head dim


// 16 rows x 16 channels += A-tile rows (16 x k) . V^T-tile rows (16 channels x k) over the k range [KB, KB + KLEN) of a chunk.
// a: first element of this lane's tile row (row l15 of the group), b: of this lane's channel row; kg = lane >> 4.
template <typename T> struct Sweep16;
template <> struct Sweep16<bf16_t> {
  template <int KB, int KLEN>
  static __device__ __forceinline__ f32x4_acc run(const bf16_t* a, const bf16_t* b, int kg, f32x4_acc acc) {
#pragma unroll
    for (int kk = KB; kk < KB + KLEN; kk += 32) {
      const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(a + kk + 8 * kg);
      const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(b + kk + 8 * kg);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
};
template <> struct Sweep16<f16_t> {
  template <int KB, int KLEN>
  static __device__ __forceinline__ f32x4_acc run(const f16_t* a, const f16_t* b, int kg, f32x4_acc acc) {
#pragma unroll
    for (int kk = KB; kk < KB + KLEN; kk += 32) {
      const f16x8_t fa = *reinterpret_cast<const f16x8_t*>(a + kk + 8 * kg);
      const f16x8_t fb = *reinterpret_cast<const f16x8_t*>(b + kk + 8 * kg);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
};
template <> struct Sweep16<float> {
  template <int KB, int KLEN>   // permuted k: lane group kg covers [KB + kg * KLEN / 4, + KLEN / 4)
  static __device__ __forceinline__ f32x4_acc run(const float* a, const float* b, int kg, f32x4_acc acc) {
#pragma unroll
    for (int q = 0; q < KLEN / 4; q += 4) {
      const float4 fa = *reinterpret_cast<const float4*>(a + KB + kg * (KLEN / 4) + q);
      const float4 fb = *reinterpret_cast<const float4*>(b + KB + kg * (KLEN / 4) + q);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, acc, 0, 0, 0);
    }
    return acc;
  }
};

// fp32 store type in split mode: the same sweep as bf16 hi / lo products (three v_mfma_f32_16x16x32_bf16 per 32 k instead of
// eight v_mfma_f32_16x16x4_f32 at a sixteenth of the rate: 48 vs 256 matrix-core cycles), both operands split into hi / lo while
// their fragments are loaded (8 consecutive k per lane: two 16-byte LDS reads; 16 lanes = 16 rows at the fp32 pitch cover all
// banks), like the scores and like the resident K8 kernel's P.V.  NR row groups x 2 channel groups per call.
template <int NR, int KB, int KLEN>
__device__ __forceinline__ void sweep_split_f32(const float* a, int a_step, const float* b, int b_step, int kg, f32x4_acc (*acc)[2]) {
#pragma unroll
  for (int kk = KB; kk < KB + KLEN; kk += 32) {
    bf16x8_t ah[NR], al[NR], bh[2], bl[2];
    auto frag = [&](const float* p, bf16x8_t* hi, bf16x8_t* lo) __attribute__((always_inline)) {
      const float4 u = *reinterpret_cast<const float4*>(p + kk + 8 * kg), v = *reinterpret_cast<const float4*>(p + kk + 8 * kg + 4);
      bf16x4_t h0, l0, h1, l1;
      split4(u, &h0, &l0);
      split4(v, &h1, &l1);
      *hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      *lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int hr = 0; hr < NR; ++hr) frag(a + hr * a_step, &ah[hr], &al[hr]);
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) frag(b + cg * b_step, &bh[cg], &bl[cg]);
#pragma unroll
    for (int hr = 0; hr < NR; ++hr)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        acc[hr][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[hr], bl[cg], acc[hr][cg], 0, 0, 0);
        acc[hr][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[hr], bh[cg], acc[hr][cg], 0, 0, 0);
        acc[hr][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[hr], bh[cg], acc[hr][cg], 0, 0, 0);
      }
  }
}

// Key rows of the frame as bf16 hi / lo MFMA fragments in FRAGMENT-MAJOR order, once per launch (QK == 2).
// Why.  Ablation builds of the statistics pass at 1024^2 (N = 4096; whole launch 231 us): without the (max, exp, sum)
// chain 233 us, without the score MFMAs 228 us, without the rel-pos LDS reads 231 us, WITHOUT THE KEY LOADS 176 us; four
// fragment register sets instead of two (three chunks in flight): 231 us.  The pass is bound by the key loads, and not by
// their latency: a lane owns one key row of the chunk (the MFMA's A-operand row), so neighbouring lanes read 16-byte
// pieces 256 bytes apart and the texture addresser works through a wave's dwordx4 load piece by piece (~64 cycles per
// instruction; 4 per chunk and wave, 8 waves per CU: 2048 of the 1840 cycles a chunk took).  Here every (16-key block, k-block m, hi | lo)
// is stored as the 64 lanes' 16-byte pieces in lane order: a wave's load instruction reads 1 KB of consecutive bytes.
// The conversion of a key row, repeated by every row tile before (128 x at 1024^2; ~50 of the pass's 177 VALU
// instructions per chunk and wave), happens once.  Layout: uint4 index (((bh * NKB + key / 16) * 2 + m) * 2 + hl) * 64 +
// kg * 16 + key % 16, NKB = ceil(N / 16); keys past N are zero rows.  With a rel-pos key grid every grid row starts a new
// 16-key block (evt_key_blocks, evt_prep_roles.h: NKB = gh ceil(gw / 16), zero rows behind a grid row's last key).
// One workgroup per (clip, 16-key block, group of 4 heads): the block's key rows are read as 1 KB token-row slices
// (coalesced), converted, placed at their fragment positions in LDS and written out as 4 KB of consecutive bytes per head
// (one thread per 8 channels with two scattered 16-byte stores each: 11.2 us at 1024^2; one workgroup for all 12 heads of a
// block: 111 workgroups at 672^2, 7.5 us).
constexpr int SKH = 4;   // heads per workgroup
__global__ __launch_bounds__(256) void split_keys_kernel(const float* __restrict__ ksrc, int64_t k_rs, int k_off, uint4* __restrict__ out, int B, int H, int Nk, int NKB, int gw) {
  __shared__ __attribute__((aligned(16))) uint4 tile[SKH * 256];   // [head][256]: a head's 4 KB block in its final order
  evt_split_keys_role(ksrc, k_rs, k_off, out, B, H, Nk, NKB, gw, (int)blockIdx.x, (int)blockIdx.y, tile);   // evt_prep_roles.h
}

// Rel-pos terms of a tile's rows in LDS, per row: [0, gh) row terms ty, -inf up to TYP = (gh + 4) & ~3 (slot gh is what
// key blocks past the grid read), then from TYP the column terms tx of 16 ceil(gw / 16) columns, -inf behind column gw (the zero
// rows of the key plane are masked by their term).  Pitch = 4 x odd words: the 16-byte reads of tx (16 lanes = 16 rows, same
// column quad) cover all 64 banks, and so do the 4-byte reads of 16 rows.
__host__ __device__ inline int stream_rel_typ(int gh) { return (gh + 4) & ~3; }
__host__ __device__ inline int stream_rel_pitch(int gh, int gw) {
  if (gw <= 0) return 1;
  const int q = (stream_rel_typ(gh) + 16 * ((gw + 15) >> 4)) >> 2;
  return 4 * (q | 1);
}

// QK: 1 = exact fp32 products (v_mfma_f32_16x16x4_f32), 2 = q, k as bf16 hi + lo (three v_mfma_f32_16x16x32_bf16 per
// product, ~1e-5 relative; the arithmetic of evt_qk's split mode).  FIRST: first frame of a clip (see the header).
// NHR: MFMA row groups of 16 query rows per workgroup (tile height 16 NHR).
template <typename T, bool FIRST, int QK, int NHR>
__global__ __launch_bounds__(256, 2) void attn_stream_kernel(const StreamArgs a, int tiles_x, int tiles_total) {
  constexpr int P = Tile<T>::PITCH, FRT = 16 * NHR;
  static_assert((size_t)2 * FRT * SDH * sizeof(float) <= (size_t)(2 * FRT + 2 * SDH) * P * sizeof(T), "the epilogue tiles alias the chunk tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char stream_smem[];
  T* An = reinterpret_cast<T*>(stream_smem);            // [FRT][P] a~ (first frame: a) of the chunk
  T* Ad = An + FRT * P;                                 // [FRT][P] da~
  T* Vd = Ad + FRT * P;                                 // [64][P]  dv~^T (first frame: v^T) of the chunk
  T* Vo = Vd + SDH * P;                                 // [64][P]  v_old^T
  const int nrel = a.gh + a.gw, RP = stream_rel_pitch(a.gh, a.gw), TYP = stream_rel_typ(a.gh);
  const int KBR = (a.gw + 15) >> 4;                     // key blocks per grid row (key plane, rel-pos grid)
  float* relv = reinterpret_cast<float*>(Vo + SDH * P); // [FRT][RP] rel-pos terms of the tile's rows (stream_rel_pitch)
  float* wst = relv + FRT * RP;                         // [4][FRT][2] per-wave (max, sum) of each row
  float* fin = wst + 4 * FRT * 2;                       // [FRT][2] row max, 1 / row sum
  float* red1 = reinterpret_cast<float*>(stream_smem);  // [FRT][64] epilogue (the chunk tiles are idle then)
  float* red2 = red1 + FRT * SDH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware placement (workgroup w runs on XCD w % 8): each XCD owns a contiguous run of (head, row tile) pairs, so
  // the K rows of one head are fetched into one or two private L2s instead of all eight.
  int t;
  {
    const int w = blockIdx.x, x = w & 7, sidx = w >> 3, q8 = tiles_total / 8, r8 = tiles_total % 8;
    t = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + sidx;
  }
  const int bh = t / tiles_x, tile_x = t - bh * tiles_x;
  const int b = bh / a.H, h = bh - b * a.H;
  const int i0 = tile_x * FRT;
#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == 100 && wave == 0;
  unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif
  const bool rel = a.rel_terms != nullptr;
  const float inv_gw = rel ? 1.0f / (float)a.gw : 0.f;
  const int l15 = lane & 15, kg = lane >> 4;
  const int64_t rs = 3 * (int64_t)a.D;
  const float* clip = a.qkv + (int64_t)b * a.N * rs;
  const float* kclip = a.ksrc + (int64_t)b * a.Nk * a.k_rs + a.k_off;
  T* stT = reinterpret_cast<T*>(a.a_state_t) + (int64_t)bh * a.Nk * a.N;   // [key j][row i]

  // channel of float4 piece p (0..3) of this lane (see evt_attn_fused.hip): QK == 1, 16x16x4 fp32 tiles with k
  // permuted, 16 contiguous channels 16 kg + 4 p; QK == 2, 16x16x32 bf16 tiles, k-block p >> 1 holds 32 (p >> 1) + 8 kg .. + 8
  auto chan = [&](int p_) __attribute__((always_inline)) { return QK == 1 ? 16 * kg + 4 * p_ : 32 * (p_ >> 1) + 8 * kg + 4 * (p_ & 1); };
  auto split8 = [&](const float4 u, const float4 v, bf16x8_t* hi, bf16x8_t* lo) __attribute__((always_inline)) {
    bf16x4_t h0, l0, h1, l1;
    split4(u, &h0, &l0);
    split4(v, &h1, &l1);
    *hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  // ---- q rows of the tile -> MFMA fragments (kept for both passes); rel-pos terms -> LDS -------------------------
  float4 qf[NHR][4];
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) {
    const int i = i0 + hr * 16 + l15;
    const float* qp = clip + (int64_t)(i < a.N ? i : a.N - 1) * rs + h * SDH;
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_) qf[hr][p_] = *reinterpret_cast<const float4*>(qp + chan(p_));
  }
  // Key fragments -> 4 pieces.  QK == 1: fp32 channels of key row j from the token buffer.  QK == 2: pieces 2m, 2m + 1 = hi,
  // lo of k-block m from the fragment-major plane (split_keys_kernel).
  const int NKB = evt_key_blocks(a.Nk, a.gh, a.gw);
  const uint4* ksp = reinterpret_cast<const uint4*>(a.k_split) + (int64_t)bh * NKB * 256;
  auto load_kf = [&](int j, float4* kf) __attribute__((always_inline)) {   // key row j (caller clamps): a gather
    if (QK == 2) {
      int pos = j;   // slot of key j in the plane: grid rows start new blocks
      if (rel) { const int ky = fast_div(j, inv_gw); pos = (int)evt_mul24(ky, KBR * 16) + (j - (int)evt_mul24(ky, a.gw)); }
      const uint4* kp = ksp + (int64_t)(pos >> 4) * 256 + kg * 16 + (pos & 15);
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_) kf[p_] = __builtin_bit_cast(float4, kp[64 * p_]);
    } else {
      const float* kp = kclip + (int64_t)j * a.k_rs + h * SDH;
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_) kf[p_] = *reinterpret_cast<const float4*>(kp + chan(p_));
    }
  };
  auto load_chunk = [&](int c0, float4* kf) __attribute__((always_inline)) {   // slots c0 + 16 wave + l15 of a streamed chunk (clamped past the end)
    if (QK == 2) {   // the wave's 16-key block of the plane: four loads of 1 KB of consecutive bytes
      const int kb = (c0 >> 4) + wave;
      const uint4* kp = ksp + (int64_t)(kb < NKB ? kb : NKB - 1) * 256 + lane;
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_) kf[p_] = __builtin_bit_cast(float4, kp[64 * p_]);
    } else {
      const int j = c0 + wave * 16 + l15;
      load_kf(j < a.Nk ? j : a.Nk - 1, kf);
    }
  };
  float4 kA[4], kB[4];
  load_chunk(0, kA);
  load_chunk(64, kB);
  if (rel) {
    // the tile's rows are contiguous in rel_terms: a flat copy, 8 loads in flight per thread before the first LDS store
    // (a loop of load -> store pairs is 16-24 serialised round trips: 12 us of a 60 us workgroup)
    const float* src = a.rel_terms + ((int64_t)bh * a.N + i0) * nrel;
    const int total = min(FRT, a.N - i0) * nrel;
    const float inv_nrel = 1.0f / (float)nrel;
    for (int e0 = 0; e0 < total; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + tid + 256 * u; v[u] = src[e < total ? e : total - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + tid + 256 * u;
        if (e < total) { const int r = fast_div(e, inv_nrel), c = e - r * nrel; relv[r * RP + (c < a.gh ? c : TYP + c - a.gh)] = v[u]; }
      }
    }
    const int npad = RP - nrel, ypad = TYP - a.gh;   // the -inf slots of every row
    const float inv_npad = 1.0f / (float)npad;
    for (int e = tid; e < FRT * npad; e += 256) {
      const int r = fast_div(e, inv_npad), c = e - r * npad;
      relv[r * RP + (c < ypad ? a.gh + c : TYP + a.gw + c - ypad)] = -INFINITY;
    }
  }
  {
    const float inv = 1.0f / a.scale;
    const bool pow2 = (inv * a.scale == 1.0f) && ((__float_as_uint(a.scale) & 0x007fffffu) == 0u);
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_) {   // q / self.scale (blocks.py:514); a power-of-two scale: exact multiply
        float4 q = qf[hr][p_];
        if (pow2) { q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv; }
        else { q.x /= a.scale; q.y /= a.scale; q.z /= a.scale; q.w /= a.scale; }
        qf[hr][p_] = q;
      }
  }
  bf16x8_t qh[NHR][2], ql[NHR][2];
  if (QK == 2) {
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int m = 0; m < 2; ++m) split8(qf[hr][2 * m], qf[hr][2 * m + 1], &qh[hr][m], &ql[hr][m]);
  }
  // k (q / scale)^T: the 16 keys whose fragments are in kf against the tile's rows.  Keys are the A operand: sacc[hr][r] is
  // the score of query row 16 hr + l15 and key 4 kg + r of the 16.
  // `preset`: the accumulators already hold the rel-pos terms (split mode: the terms are the products' starting value)
  auto scores = [&](const float4* kf, f32x4_acc* sacc, const bool preset = false) __attribute__((always_inline)) {
    if (!preset) {
#pragma unroll
      for (int hr = 0; hr < NHR; ++hr) sacc[hr] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};
    }
    if (QK == 1) {
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
        for (int hr = 0; hr < NHR; ++hr) {
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].x, qf[hr][p_].x, sacc[hr], 0, 0, 0);
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].y, qf[hr][p_].y, sacc[hr], 0, 0, 0);
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].z, qf[hr][p_].z, sacc[hr], 0, 0, 0);
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[p_].w, qf[hr][p_].w, sacc[hr], 0, 0, 0);
        }
    } else {
      // (Separate accumulators for the two channel halves, 2 NHR independent chains of three MFMAs instead of NHR of six, measured
      // SLOWER: 1185 vs 1133 us for eight 1024^2 streams -- dependent MFMAs on one accumulator forward their result.)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const bf16x8_t kh = __builtin_bit_cast(bf16x8_t, kf[2 * m]), kl = __builtin_bit_cast(bf16x8_t, kf[2 * m + 1]);
#pragma unroll
        for (int hr = 0; hr < NHR; ++hr) {
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[hr][m], sacc[hr], 0, 0, 0);
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[hr][m], sacc[hr], 0, 0, 0);
          sacc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[hr][m], sacc[hr], 0, 0, 0);
        }
      }
    }
  };
  // Scores of the 16 keys in kf + the rel-pos terms of the lane's 4 keys js[0..3] (valid key indices).  Exact mode: the
  // reference's order (x + ty) + tx (utils.py:166-172).  Split mode: ty + tx is the accumulators' starting value, in every
  // pass alike (one add per score instead of two and no zeroing; the products' own rounding is ~1e-5 of the score).
  auto scores_rel = [&](const float4* kf, f32x4_acc* sacc, const int* js) __attribute__((always_inline)) {
    if (!rel) { scores(kf, sacc); return; }
    int oy[4], ox[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ky = fast_div(js[r], inv_gw);
      oy[r] = ky;
      ox[r] = TYP + js[r] - (int)evt_mul24(ky, a.gw);
    }
    if (QK == 2) {
#pragma unroll
      for (int hr = 0; hr < NHR; ++hr) {
        const float* rv = relv + (16 * hr + l15) * RP;
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[hr][r] = rv[oy[r]] + rv[ox[r]];
      }
      scores(kf, sacc, true);
      return;
    }
    scores(kf, sacc);
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr) {
      const float* rv = relv + (16 * hr + l15) * RP;
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[hr][r] = (sacc[hr][r] + rv[oy[r]]) + rv[ox[r]];
    }
  };
  __syncthreads();   // rel-pos tile
  STR_TICK(0);   // prologue: q rows, first key fragments requested, rel-pos tile

  // ---- pass A: running (max, sum) of every row over all N keys; wave w owns keys 16 w .. 16 w + 15 of each 64-key chunk ----
  float rm[NHR], rsum[NHR];
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) { rm[hr] = -1.0e30f; rsum[hr] = 0.f; }
  // The pass is VALU-bound once the key loads are coalesced (scripts/isa_loops.py: 127 VALU per chunk and wave), so the
  // per-key bookkeeping is kept out of it: the (grid row, grid column) of the lane's first key advance by one chunk's
  // 64 keys per call instead of a float division per key, and only the chunks that can reach past N mask their scores.
  const int cdy = rel ? 64 / a.gw : 0, cdx = rel ? 64 - cdy * a.gw : 0;   // a chunk = cdy grid rows + cdx columns
  int ky0 = 0, kx0 = 0;                                                    // of key 16 wave + 4 kg of the current chunk
  if (rel) { const int j0 = wave * 16 + 4 * kg; ky0 = fast_div(j0, inv_gw); kx0 = j0 - ky0 * a.gw; }
  constexpr float L2E = 1.44269504088896340736f;
  // (max, sum) update of the tile's rows by a chunk's scores, 4 scores of ONE row per row group: one rescale of the running sum
  // per 4 scores, one exponential per score, everything in log2 units (ref2 = running maximum x log2e) so that the subtraction of
  // the maximum is the exponent's fma.  Measured and not kept: a lazily raised REFERENCE instead of the exact maximum (raised only
  // when a score exceeds it by more than 8: no fifth exponential, 40 instead of 52 VALU instructions per chunk and wave, one
  // wave-uniform branch) -- within noise, the pass is not bound by its VALU instruction count (1024^2, eight streams: 1201 vs 1185 us).
  float ref2[NHR];
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) ref2[hr] = rm[hr] * L2E;
  auto update = [&](const float (*x)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr) {
      const float nref = fmaxf(ref2[hr], fmaxf(fmaxf(x[hr][0], x[hr][1]), fmaxf(x[hr][2], x[hr][3])) * L2E);   // finite: starts at -1e30 log2e
      const float part = (__builtin_amdgcn_exp2f(fmaf(x[hr][0], L2E, -nref)) + __builtin_amdgcn_exp2f(fmaf(x[hr][1], L2E, -nref))) +
                         (__builtin_amdgcn_exp2f(fmaf(x[hr][2], L2E, -nref)) + __builtin_amdgcn_exp2f(fmaf(x[hr][3], L2E, -nref)));
      rsum[hr] = fmaf(rsum[hr], __builtin_amdgcn_exp2f(ref2[hr] - nref), part);
      ref2[hr] = nref;
    }
  };
  // Split mode with a rel-pos grid: the wave's 16-key block lies inside ONE grid row of the key plane -- its grid row gky and
  // column block gxb are wave-uniform (scalar registers, advanced by 4 blocks per chunk), the row term is one LDS word and the
  // 4 column terms one 16-byte read per row group, they are the accumulators' starting value, and the plane's zero rows and the
  // blocks past the grid are masked by -inf terms: 2 LDS reads + 4 adds per row group instead of 8 reads, 8 adds, the index
  // arithmetic and the masks.
  int gky = 0, gxb = 0, sdy = 0, sdx = 0;
  if (QK == 2 && rel) { gky = wave / KBR; gxb = wave - gky * KBR; sdy = 4 / KBR; sdx = 4 - sdy * KBR; }
  // The terms of a chunk are REQUESTED A CHUNK AHEAD (behind the score MFMAs' issue, in front of the previous chunk's (max, exp,
  // sum) instructions): they are the accumulators' starting value, i.e. the first thing a chunk needs, and an LDS round trip
  // in front of every chunk's MFMAs is not hidden by two waves per SIMD (1-2 % of the launch).
  float nty[NHR];
  float4 ntx[NHR];
  auto request_terms = [&]() __attribute__((always_inline)) {
    const int tyo = min(gky, a.gh), txo = TYP + 16 * gxb + 4 * kg;
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr) {
      const float* rv = relv + (16 * hr + l15) * RP;
      nty[hr] = rv[tyo];
      ntx[hr] = *reinterpret_cast<const float4*>(rv + txo);
    }
    gxb += sdx;
    gky += sdy;
    if (gxb >= KBR) { gxb -= KBR; ++gky; }
  };
  if (QK == 2 && rel) request_terms();
  auto stats_grid = [&](const float4* kf) __attribute__((always_inline)) {
    f32x4_acc sacc[NHR];
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr) sacc[hr] = (f32x4_acc){nty[hr] + ntx[hr].x, nty[hr] + ntx[hr].y, nty[hr] + ntx[hr].z, nty[hr] + ntx[hr].w};
    scores(kf, sacc, true);
    request_terms();
    float x[NHR][4];
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[hr][r] = sacc[hr][r];
    update(x);
  };
  auto stats = [&](int c0, const float4* kf, const bool mask) __attribute__((always_inline)) {
    if (QK == 2 && rel) { stats_grid(kf); return; }
    f32x4_acc sacc[NHR];
    scores(kf, sacc);
    const int jb = c0 + wave * 16 + 4 * kg;
    if (rel) {
      int oy[4], ox[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kx = kx0 + r;
        const bool wrap = kx >= a.gw;
        oy[r] = min(ky0 + (wrap ? 1 : 0), a.gh - 1);   // (keys past N: any valid slot, their scores are masked)
        ox[r] = TYP + (wrap ? kx - a.gw : kx);
      }
#pragma unroll
      for (int hr = 0; hr < NHR; ++hr) {
        const float* rv = relv + (16 * hr + l15) * RP;
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[hr][r] = (sacc[hr][r] + rv[oy[r]]) + rv[ox[r]];
      }
      kx0 += cdx;
      ky0 += cdy;
      if (kx0 >= a.gw) { kx0 -= a.gw; ++ky0; }
    }
    float x[NHR][4];
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[hr][r] = (!mask || jb + r < a.Nk) ? sacc[hr][r] : -INFINITY;
    update(x);
  };
  // two fragment register sets used alternately, every load unconditional (clamped key): the waits can then count on the
  // younger requests being in flight (a copy kf = kn at the loop's back edge makes hipcc wait for the prefetch just issued)
  {
    const int NS = (QK == 2 && rel) ? NKB * 16 : a.Nk;  // slots to stream (the key plane's, with a rel-pos grid)
    int c0 = 0;
    for (; c0 + 128 <= NS; c0 += 128) {   // whole chunk pairs: no masks
      stats(c0, kA, false);
      load_chunk(c0 + 128, kA);
      stats(c0 + 64, kB, false);
      load_chunk(c0 + 192, kB);
    }
    if (c0 < NS) stats(c0, kA, true);          // the last (partial) pair: its fragments were requested above / in the prologue
    if (c0 + 64 < NS) stats(c0 + 64, kB, true);
  }
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) {
    // The combine works in natural units.  The product is PINNED: contracted into the combine's "m - M" as fma(ref2, ln2, -M), with
    // M the ROUNDED product of a lane that has seen no key (reference -1e30: the never-valid key slots of a narrow grid), the
    // difference is the product's rounding error, ~1e22, and exp2 of it is inf or 0 -- 0 x inf = NaN in every row of the tile.
    float v = ref2[hr] * 0.69314718055994530942f;
    asm volatile("" : "+v"(v));
    rm[hr] = v;
  }
  bool rok[NHR];   // query row 16 hr + l15 of the tile exists
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) rok[hr] = i0 + 16 * hr + l15 < a.N;
  // ---- gated frame: selected-column bookkeeping.  The first two chunks' column indices, and the first chunk's key rows
  // and old reference values, are requested HERE -- they do not depend on the statistics -- so that their two dependent
  // round trips overlap the combine below.
  const int cnt = FIRST ? 0 : (a.count ? a.count[b] : a.kcap);
  const int32_t* ix = a.idx + (int64_t)b * a.kcap;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int VIT = SDH * (FKC / VEC) / 256;   // 16-byte pieces of dv~^T / v_old^T per thread and chunk
  const T* Vg_d = reinterpret_cast<const T*>(a.v_delta_t) + (int64_t)bh * SDH * a.kcap;
  const T* Vg_o = reinterpret_cast<const T*>(a.v_old_t) + (int64_t)bh * SDH * a.kcap;
  const bool vvec = (a.kcap % VEC) == 0;
  // The chain index -> key row of the selected column -> scores -> gate is latency-bound (two dependent round trips per
  // chunk): the column indices are fetched two chunks ahead and the key rows + old reference values one chunk ahead,
  // into two register sets used alternately (no register copies of loaded values at the loop's back edge).
  struct Cols { int frag; int g[4]; };   // this lane's fragment column (key row it loads) and its 4 gate columns; -1 past the count
  auto cols_of = [&](int c) __attribute__((always_inline)) {
    Cols q;
    const int kf_ = c * FKC + wave * 16 + l15;
    const int jf = ix[kf_ < cnt ? kf_ : 0];
    q.frag = kf_ < cnt ? jf : -1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = c * FKC + wave * 16 + 4 * kg + r;
      const int j = ix[kk < cnt ? kk : 0];
      q.g[r] = kk < cnt ? j : -1;
    }
    return q;
  };
  // element (key j, row i) of this head's reference: a 32-bit byte offset from the head's (wave-uniform) base -- the launcher
  // checks N^2 sizeof(T) < 2^32 -- so that an access is one 24-bit multiply, one add and a scalar-base load / store instead of
  // 64-bit address arithmetic per element
  auto st_at = [&](uint32_t col_off, int i) __attribute__((always_inline)) -> T* {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(stT) + (col_off + (uint32_t)i * (uint32_t)sizeof(T)));
  };
  auto request_old = [&](const Cols& q, T (*old)[4]) __attribute__((always_inline)) {   // old reference values of the lane's 4 columns
    uint32_t co[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) co[r] = evt_mul24(q.g[r] >= 0 ? q.g[r] : 0, a.N * (int)sizeof(T));
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr) {
      const int i = rok[hr] ? i0 + 16 * hr + l15 : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        old[hr][r] = *st_at(co[r], i);
      }
    }
  };
  auto load_v = [&](int k0, uint4* pd, uint4* po) __attribute__((always_inline)) {   // chunk k0 of dv~^T / v_old^T -> registers (branch-free, clamped)
    // columns in [count, kcap) hold zeros (evt_v_gate writes them); pieces past kcap are zeroed here
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC, kv = k0 + jj;
      const bool in = kv < a.kcap;
      const int64_t o = (int64_t)d * a.kcap + (in ? kv : 0);
      const uint4 xd = *reinterpret_cast<const uint4*>(Vg_d + o), xo = *reinterpret_cast<const uint4*>(Vg_o + o);
      pd[it] = in ? xd : make_uint4(0, 0, 0, 0);
      po[it] = in ? xo : make_uint4(0, 0, 0, 0);
    }
  };
  constexpr bool VPRE = sizeof(T) == 2;   // 16-bit store: V pieces requested a whole chunk ahead (see the chunk loop)
  uint4 vnd[VPRE ? VIT : 1], vno[VPRE ? VIT : 1];
  const int nch = FIRST ? 0 : (cnt + FKC - 1) / FKC;
  T old[NHR][4];   // ONE set: the next chunk's values are requested as soon as the gate has consumed this chunk's
  Cols qA, qB;
  uint4 v0d[VPRE ? 1 : VIT], v0o[VPRE ? 1 : VIT];   // fp32: the first chunk's V pieces, in registers across the combine only
  if (!FIRST && nch > 0) {
    qA = cols_of(0);
    qB = cols_of(1);
    load_kf(qA.frag >= 0 ? qA.frag : 0, kA);
    request_old(qA, old);
    if (vvec) {
      if (VPRE) load_v(0, vnd, vno);
      else load_v(0, v0d, v0o);
    }
  }
  STR_TICK(1);   // pass A
  // combine the 4 key groups of a row (lanes l15, l15 + 16, + 32, + 48), then the 4 waves through LDS
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) {
    float m = rm[hr], s = rsum[hr];
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
      const float M = fmaxf(m, m2);
      s = s * fast_exp(m - M) + s2 * fast_exp(m2 - M);
      m = M;
    }
    if (kg == 0) {
      wst[(wave * FRT + 16 * hr + l15) * 2] = m;
      wst[(wave * FRT + 16 * hr + l15) * 2 + 1] = s;
    }
  }
  __syncthreads();
  if (tid < FRT) {
    float m = wst[tid * 2], s = wst[tid * 2 + 1];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float m2 = wst[(w * FRT + tid) * 2], s2 = wst[(w * FRT + tid) * 2 + 1];
      const float M = fmaxf(m, m2);
      s = s * fast_exp(m - M) + s2 * fast_exp(m2 - M);
      m = M;
    }
    fin[tid * 2] = m;
    fin[tid * 2 + 1] = 1.0f / s;   // softmax normaliser as a reciprocal: e * (1 / sum) is within 1 ulp of e / sum before the rounding
  }
  __syncthreads();
  float fm[NHR], fi[NHR];
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr) {
    fm[hr] = fin[(16 * hr + l15) * 2];
    fi[hr] = fin[(16 * hr + l15) * 2 + 1];
  }
  if (!FIRST && !VPRE && vvec && nch > 0) {   // fp32: the first chunk's V pieces -> LDS (the V tiles are idle until the chunk loop)
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC;
      *reinterpret_cast<uint4*>(Vd + d * P + jj) = v0d[VPRE ? 0 : it];
      *reinterpret_cast<uint4*>(Vo + d * P + jj) = v0o[VPRE ? 0 : it];
    }
  }
  STR_TICK(2);   // statistics combine

  // ---- pass B ----------------------------------------------------------------------------------------------------
  const int half = wave & 1, psel = wave >> 1;   // gated: product (0: a~ . dv~, 1: da~ . v_old); first frame: key half of the chunk
  f32x4_acc acc[NHR][2];   // rows 16 hr + 4 kg + r, channel 32 half + 16 cg + l15
#pragma unroll
  for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) acc[hr][cg] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};
  T* pv = reinterpret_cast<T*>(a.pv);
  // epilogue pieces of this thread: 8 channels of one row each
  constexpr int PIT = (FRT * 8 + 255) / 256;
  union Pv8 { uint4 u[(8 * sizeof(T)) / 16]; T t[8]; };
  union Ref8 { float4 v[2]; float f[8]; };
  Pv8 pvr[PIT];
  Ref8 nrr[PIT];
  auto piece = [&](int it, int* row, int* c8, int64_t* off) __attribute__((always_inline)) {
    const int e = tid + 256 * it, rr = e >> 3, i = i0 + rr;
    *row = rr;
    *c8 = (e & 7) * 8;
    *off = ((int64_t)b * a.N + (i < a.N ? i : a.N - 1)) * a.D + h * SDH + (e & 7) * 8;
    return e < FRT * 8 && i < a.N;
  };
  typedef Quad<T> QuadT;
  // both accumulator products of the chunk staged in LDS: rows of a~ / da~ (A operand) x channels of dv~^T / v_old^T (B)
  constexpr bool SWEEP_SPLIT = QK == 2 && sizeof(T) == 4;   // fp32 store type, split arithmetic: bf16 hi / lo sweeps
  auto sweep = [&](const T* At, const T* Vt) __attribute__((always_inline)) {
    if constexpr (SWEEP_SPLIT) {
      sweep_split_f32<NHR, 0, FKC>(reinterpret_cast<const float*>(At) + l15 * P, 16 * P, reinterpret_cast<const float*>(Vt) + (32 * half + l15) * P, 16 * P, kg, acc);
      return;
    }
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg)
        acc[hr][cg] = Sweep16<T>::template run<0, FKC>(At + (16 * hr + l15) * P, Vt + (32 * half + 16 * cg + l15) * P, kg, acc[hr][cg]);
  };

  if (FIRST) {
    const T* vst = reinterpret_cast<const T*>(a.v_state) + (int64_t)b * a.Nk * a.D + h * SDH;
    // keys in index order here (the state columns and value rows are): key c0 + 16 wave + l15, gathered from the plane
    auto load_keys = [&](int c0, float4* kf) __attribute__((always_inline)) { const int j = c0 + wave * 16 + l15; load_kf(j < a.Nk ? j : a.Nk - 1, kf); };
    load_keys(0, kA);
    load_keys(64, kB);
    auto first_chunk = [&](int c0, const float4* kf) __attribute__((always_inline)) {
      // the chunk's 64 value rows (16 channels per thread), requested ahead of the score MFMAs
      const int vkey = tid >> 2, vc0 = (tid & 3) * 16, vj = c0 + vkey;
      union { uint4 u[(16 * sizeof(T)) / 16]; T t[16]; } vv;
#pragma unroll
      for (int q = 0; q < (int)((16 * sizeof(T)) / 16); ++q)
        vv.u[q] = reinterpret_cast<const uint4*>(vst + (int64_t)(vj < a.Nk ? vj : a.Nk - 1) * a.D + vc0)[q];
      f32x4_acc sacc[NHR];
      const int jb = c0 + wave * 16 + 4 * kg;
      int js[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) js[r] = jb + r < a.Nk ? jb + r : a.Nk - 1;
      scores_rel(kf, sacc, js);
#pragma unroll
      for (int hr = 0; hr < NHR; ++hr) {
        QuadT nw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = rok[hr] && jb + r < a.Nk;
          const float an = Store<T>::round(fast_exp(sacc[hr][r] - fm[hr]) * fi[hr]);
          Store<T>::store(&nw.t[r], ok ? an : 0.f);
          if (ok) Store<T>::store(stT + (int64_t)(jb + r) * a.N + i0 + 16 * hr + l15, an);
        }
        *reinterpret_cast<decltype(nw.v)*>(An + (16 * hr + l15) * P + wave * 16 + 4 * kg) = nw.v;
      }
      T z;
      Store<T>::store(&z, 0.f);
#pragma unroll
      for (int q = 0; q < 16; ++q) Vd[(vc0 + q) * P + vkey] = vj < a.Nk ? vv.t[q] : z;
      __syncthreads();
      // keys 0..31 of the chunk on waves 0, 1, keys 32..63 on waves 2, 3 (summed unrounded in the epilogue)
      if constexpr (SWEEP_SPLIT) {
        const float* At = reinterpret_cast<const float*>(An) + l15 * P;
        const float* Vt = reinterpret_cast<const float*>(Vd) + (32 * half + l15) * P;
        if (psel == 0) sweep_split_f32<NHR, 0, 32>(At, 16 * P, Vt, 16 * P, kg, acc);
        else sweep_split_f32<NHR, 32, 32>(At, 16 * P, Vt, 16 * P, kg, acc);
      } else
#pragma unroll
      for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
          const T* At = An + (16 * hr + l15) * P;
          const T* Vt = Vd + (32 * half + 16 * cg + l15) * P;
          if (psel == 0) acc[hr][cg] = Sweep16<T>::template run<0, 32>(At, Vt, kg, acc[hr][cg]);
          else acc[hr][cg] = Sweep16<T>::template run<32, 32>(At, Vt, kg, acc[hr][cg]);
        }
      __syncthreads();
    };
    for (int c0 = 0; c0 < a.Nk; c0 += 128) {
      first_chunk(c0, kA);
      load_keys(c0 + 128, kA);
      if (c0 + 64 < a.Nk) first_chunk(c0 + 64, kB);
      load_keys(c0 + 192, kB);
    }
    STR_TICK(3);   // pass B (first frame: all keys)
    // ---- epilogue: out = state = round(acc(keys 0..31 of each chunk) + acc(keys 32..63)) ------------------------
    float* redp = (psel == 0) ? red1 : red2;
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int r = 0; r < 4; ++r) redp[(16 * hr + 4 * kg + r) * SDH + 32 * half + 16 * cg + l15] = acc[hr][cg][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      int row, c8; int64_t eo;
      if (!piece(it, &row, &c8, &eo)) continue;
      Pv8 st8;
      Ref8 o8;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v = Store<T>::round(red1[row * SDH + c8 + q] + red2[row * SDH + c8 + q]);
        Store<T>::store(&st8.t[q], v);
        o8.f[q] = v;
      }
#pragma unroll
      for (int q = 0; q < (int)((8 * sizeof(T)) / 16); ++q) reinterpret_cast<uint4*>(pv + eo)[q] = st8.u[q];
      if (a.out_f32 != nullptr) {
        reinterpret_cast<float4*>(a.out_f32 + eo)[0] = o8.v[0];
        reinterpret_cast<float4*>(a.out_f32 + eo)[1] = o8.v[1];
      }
    }
#ifdef EVT_PROF
    STR_TICK(4);
    if (prof_on && lane == 0)
      for (int q = 0; q < 16; ++q) evt_prof_stream_buf[q] = prof_acc[q];
#endif
    return;
  }

  // ---- gated frame: the k selected columns in chunks of 64 (declarations: in front of the statistics combine) ----
  // `full` (wave-uniform): every row of the tile and every column of the chunk exists -- the reference stores are then
  // unconditional.  (Predicated stores sit in exec-masked branches; hipcc cannot count them and waits vmcnt(0) at the next
  // counted wait, i.e. for the prefetched loads of the NEXT chunk and for the stores' own acknowledgements.)
  const bool tile_full = i0 + FRT <= a.N;
  auto process = [&](int k0, const Cols& q, const Cols& qnext, const float4* kf, uint4* vpd, uint4* vpo) __attribute__((always_inline)) {
    f32x4_acc sacc[NHR];
    int js[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) js[r] = q.g[r] >= 0 ? q.g[r] : 0;
    scores_rel(kf, sacc, js);
    const bool full = tile_full && k0 + FKC <= cnt;
    uint32_t co[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) co[r] = evt_mul24(js[r], a.N * (int)sizeof(T));
    T zero_t;
    Store<T>::store(&zero_t, 0.f);
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr) {
      QuadT nw, dl;   // a~ and da~ of the lane's 4 columns, rounded to the store type (modules.py:187-201), four at a time
      float e[4], an[4], d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = fast_exp(sacc[hr][r] - fm[hr]) * fi[hr];
      Store<T>::round4(e, nw.t, an);
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] = an[r] - Store<T>::load(&old[hr][r]);
      Store<T>::round4(d, dl.t, nullptr);
      if (full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *st_at(co[r], i0 + 16 * hr + l15) = nw.t[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = rok[hr] && q.g[r] >= 0;
          if (ok) *st_at(co[r], i0 + 16 * hr + l15) = nw.t[r];
          else { nw.t[r] = zero_t; dl.t[r] = zero_t; }
        }
      }
      *reinterpret_cast<decltype(nw.v)*>(An + (16 * hr + l15) * P + wave * 16 + 4 * kg) = nw.v;
      *reinterpret_cast<decltype(dl.v)*>(Ad + (16 * hr + l15) * P + wave * 16 + 4 * kg) = dl.v;
    }
    request_old(qnext, old);   // (past the last chunk: dummy re-reads of column 0)
    STR_TICK(5);   // chunk: key rows + old values waited for, scores, gate
    // V pieces.  16-bit store (VPRE): this chunk's pieces were requested a chunk ago (16 registers carried over the loop's
    // back edge); they go to LDS now and the next chunk's are requested.  fp32 (32 registers): the NEXT chunk's pieces are
    // requested here -- the score / gate registers are dead -- fly during the barrier + MFMA sweep, and are written to the
    // (then idle) V tiles behind the sweep; this chunk's pieces are already in LDS.
    uint4 lpd[VIT], lpo[VIT];
    if (vvec && VPRE) {
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC;
        *reinterpret_cast<uint4*>(Vd + d * P + jj) = vpd[it];
        *reinterpret_cast<uint4*>(Vo + d * P + jj) = vpo[it];
      }
      load_v(k0 + FKC, vpd, vpo);   // (past the last chunk: a dummy, clamped)
    } else if (vvec) {
      if (k0 + FKC < cnt) load_v(k0 + FKC, lpd, lpo);
    } else {
      for (int e = tid; e < SDH * FKC; e += 256) {
        const int d = e / FKC, jj = e - d * FKC, kv = k0 + jj;
        float vd = 0.f, vo = 0.f;
        if (kv < cnt) {
          vd = Store<T>::load(Vg_d + (int64_t)d * a.kcap + kv);
          vo = Store<T>::load(Vg_o + (int64_t)d * a.kcap + kv);
        }
        Store<T>::store(Vd + d * P + jj, vd);
        Store<T>::store(Vo + d * P + jj, vo);
      }
    }
    STR_TICK(6);   // chunk: V staging
    __syncthreads();
    STR_TICK(7);   // chunk: barrier
    sweep(psel == 0 ? An : Ad, psel == 0 ? Vd : Vo);
    STR_TICK(8);   // chunk: MFMA sweep
    __syncthreads();
    if (vvec && !VPRE && k0 + FKC < cnt) {   // (wave-uniform) the next chunk's pieces -> the V tiles nobody reads any more
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        const int e = tid + 256 * it, d = e / (FKC / VEC), jj = (e - d * (FKC / VEC)) * VEC;
        *reinterpret_cast<uint4*>(Vd + d * P + jj) = lpd[it];
        *reinterpret_cast<uint4*>(Vo + d * P + jj) = lpo[it];
      }
    }
    STR_TICK(9);   // chunk: barrier
  };
  // Issue order per chunk (vmcnt retires in order): the NEXT chunk's key rows, the column indices of the chunk after that,
  // then the chunk itself; behind its gate the next chunk's old reference values and V pieces.  Every wait leaves the
  // younger requests in flight.
  if (nch > 0) {
    for (int c = 0; c < nch; c += 2) {
      load_kf(qB.frag >= 0 ? qB.frag : 0, kB);   // chunk c + 1 (a dummy re-read of key 0 past the last chunk)
      const Cols n0 = cols_of(c + 2);
      process(c * FKC, qA, qB, kA, vnd, vno);
      qA = n0;
      if (c + 1 < nch) {
        load_kf(qA.frag >= 0 ? qA.frag : 0, kA);   // chunk c + 2
        const Cols n1 = cols_of(c + 3);
        process((c + 1) * FKC, qB, qA, kB, vnd, vno);
        qB = n1;
      }
    }
  }

  // the A.v state rows (and the next gate's reference) of the epilogue: requested here, behind the last chunk -- held
  // across the chunk loop they cost 24-32 registers (spills in the 48-row variants)
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    int row, c8; int64_t eo;
    piece(it, &row, &c8, &eo);
#pragma unroll
    for (int q = 0; q < (int)((8 * sizeof(T)) / 16); ++q) pvr[it].u[q] = reinterpret_cast<const uint4*>(pv + eo)[q];
    if (a.norm_ref != nullptr) {
      nrr[it].v[0] = reinterpret_cast<const float4*>(a.norm_ref + eo)[0];
      nrr[it].v[1] = reinterpret_cast<const float4*>(a.norm_ref + eo)[1];
    }
  }
  STR_TICK(3);   // pass B (selected columns)
  // ---- epilogue: state += round(a~ . dv~); state += round(da~ . v_old); heads merged on write; per-head ||out - ref||^2 ----
  {
    float* redp = (psel == 0) ? red1 : red2;
#pragma unroll
    for (int hr = 0; hr < NHR; ++hr)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          redp[(16 * hr + 4 * kg + r) * SDH + 32 * half + 16 * cg + l15] = Store<T>::round(acc[hr][cg][r]);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    int row, c8; int64_t eo;
    const bool ok = piece(it, &row, &c8, &eo);
    const int rrow = row < FRT ? row : 0;
    Pv8 st8 = pvr[it];
    Ref8 o8;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = Store<T>::round(Store<T>::load(&st8.t[q]) + red1[rrow * SDH + c8 + q]);   // += a~ . dv~
      v = Store<T>::round(v + red2[rrow * SDH + c8 + q]);                                   // += da~ . v_old
      Store<T>::store(&st8.t[q], v);
      o8.f[q] = v;
    }
    if (ok) {
#pragma unroll
      for (int q = 0; q < (int)((8 * sizeof(T)) / 16); ++q) reinterpret_cast<uint4*>(pv + eo)[q] = st8.u[q];
      if (a.out_f32 != nullptr) {   // NULL: the caller reads the (identical) values from the A.v state
        reinterpret_cast<float4*>(a.out_f32 + eo)[0] = o8.v[0];
        reinterpret_cast<float4*>(a.out_f32 + eo)[1] = o8.v[1];
      }
    }
    if (a.norm_parts != nullptr) {   // the 8 threads of a row are consecutive lanes: fixed butterfly order (deterministic)
      float ss = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = o8.f[q] - nrr[it].f[q]; ss = fmaf(d, d, ss); }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      if (ok && (tid & 7) == 0) a.norm_parts[((int64_t)b * a.N + i0 + row) * a.H + h] = ss;
    }
  }
#ifdef EVT_PROF
  STR_TICK(4);   // epilogue
  if (prof_on && lane == 0)
    for (int q = 0; q < 16; ++q) evt_prof_stream_buf[q] = prof_acc[q];
#endif
}

template <typename T, int NHR>
size_t stream_lds_bytes(int gh, int gw) {
  constexpr int P = Tile<T>::PITCH, FRT = 16 * NHR;
  return (size_t)(2 * FRT + 2 * SDH) * P * sizeof(T) + ((size_t)FRT * stream_rel_pitch(gh, gw) + 4 * FRT * 2 + FRT * 2) * sizeof(float);
}

template <typename T, bool FIRST, int QK, int NHR>
void launch_stream_inst(const StreamArgs& a, hipStream_t s) {
  constexpr int FRT = 16 * NHR;
  const size_t lds = stream_lds_bytes<T, NHR>(a.gh, a.gw);
  const int tiles_x = (a.N + FRT - 1) / FRT, total = tiles_x * a.B * a.H;
  EVT_ALLOW_LDS((attn_stream_kernel<T, FIRST, QK, NHR>), lds);
  if (QK == 2 && !a.k_split_ready) {
    const int nkb = evt_key_blocks(a.Nk, a.gh, a.gw);
    hipLaunchKernelGGL(split_keys_kernel, dim3((unsigned)(a.B * nkb), (unsigned)((a.H + SKH - 1) / SKH)), dim3(256), 0, s, a.ksrc,
                       (int64_t)a.k_rs, a.k_off, reinterpret_cast<uint4*>(a.k_split), a.B, a.H, a.Nk, nkb, a.gw);
  }
  hipLaunchKernelGGL((attn_stream_kernel<T, FIRST, QK, NHR>), dim3(total), dim3(256), lds, s, a, tiles_x, total);
}

// Tile height: the launch runs in ceil(workgroups / slots) rounds of (almost) equal length, slots = CUs x workgroups per
// CU (two by registers; fewer when LDS says so), and a workgroup of NHR row groups costs ~(1 + NHR) units (the key
// fragments are split once per workgroup, everything else is per row group).
template <typename T>
int stream_pick_nhr(const StreamArgs& a) {
  const int cus = evt_cu_count();
  const size_t lds[2] = {stream_lds_bytes<T, 2>(a.gh, a.gw), stream_lds_bytes<T, 3>(a.gh, a.gw)};
  int best = 2;
  int64_t best_cost = -1;
  for (int nhr = 2; nhr <= 3; ++nhr) {   // (64-row tiles, NHR = 4, spill 60-600 registers in the gated variants: not built)
    const int per_cu = (int)std::min<size_t>(2, (160 * 1024) / lds[nhr - 2]);
    if (per_cu < 1) continue;
    const int64_t wgs = (int64_t)((a.N + 16 * nhr - 1) / (16 * nhr)) * a.B * a.H, slots = (int64_t)cus * per_cu;
    const int64_t cost = ((wgs + slots - 1) / slots) * (1 + nhr) * (per_cu == 1 ? 3 : 4);   // one workgroup per CU runs ~25 % faster
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nhr; }
  }
  return best;
}

template <typename T, bool FIRST>
void launch_stream(const StreamArgs& a, int qk_split, hipStream_t s) {
  if (!qk_split) { launch_stream_inst<T, FIRST, 1, 2>(a, s); return; }   // exact fp32 products: 32-row tiles only
  switch (stream_pick_nhr<T>(a)) {
    case 3: launch_stream_inst<T, FIRST, 2, 3>(a, s); break;
    default: launch_stream_inst<T, FIRST, 2, 2>(a, s); break;
  }
}

}  // namespace

extern "C" int64_t evt_attention_stream_lds_bytes(int32_t store, int32_t gh, int32_t gw) {
  const bool rel = gh > 0 && gw > 0;
  if (!rel) gh = gw = 0;
  switch (store) {
    case EVT_F32: return (int64_t)stream_lds_bytes<float, 2>(gh, gw);
    case EVT_BF16: return (int64_t)stream_lds_bytes<bf16_t, 2>(gh, gw);
    case EVT_F16: return (int64_t)stream_lds_bytes<f16_t, 2>(gh, gw);
    default: return -1;
  }
}

extern "C" int64_t evt_attention_stream_key_blocks(int32_t N, int32_t gh, int32_t gw) {
  if (N <= 0) return -1;
  const bool rel = gh > 0 && gw > 0;
  return evt_key_blocks(N, rel ? gh : 0, rel ? gw : 0);
}

extern "C" int evt_attention_stream(const evt_attn_stream_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_attention_stream: null descriptor");
  EVT_REQUIRE(d->qkv && d->a_state_t && d->pv, EVT_ERR_BAD_ARG, "evt_attention_stream: null qkv / a_state_t / pv");
  EVT_REQUIRE(d->B >= 0 && d->H > 0 && d->N > 0 && d->D == d->H * 64, EVT_ERR_BAD_SHAPE,
              "evt_attention_stream: head dim 64 required (B=%d H=%d N=%d D=%d)", d->B, d->H, d->N, d->D);
  EVT_REQUIRE(d->scale > 0.f, EVT_ERR_BAD_ARG, "evt_attention_stream: scale must be positive");
  EVT_REQUIRE(d->N <= 32767, EVT_ERR_BAD_SHAPE, "evt_attention_stream: N=%d (32-bit byte offsets into a head's N x N reference: N <= 32767)", d->N);
  EVT_REQUIRE((d->kv == nullptr) == (d->Nk == 0), EVT_ERR_BAD_ARG, "evt_attention_stream: kv and Nk come together (pooled keys), or neither");
  EVT_REQUIRE(d->Nk >= 0 && d->Nk <= d->N, EVT_ERR_BAD_SHAPE, "evt_attention_stream: Nk=%d pooled keys of N=%d tokens", d->Nk, d->N);
  const int Nk = d->kv ? d->Nk : d->N;
  EVT_REQUIRE(d->rel_terms == nullptr || (d->gh > 0 && d->gw > 0 && d->gh * d->gw == Nk), EVT_ERR_BAD_SHAPE,
              "evt_attention_stream: rel-pos key grid %dx%d does not match the %d keys", d->gh, d->gw, Nk);
  EVT_REQUIRE((d->norm_ref == nullptr) == (d->norm_parts == nullptr), EVT_ERR_BAD_ARG, "evt_attention_stream: norm_ref / norm_parts come together");
  if (d->first) {
    EVT_REQUIRE(d->v_state != nullptr, EVT_ERR_BAD_ARG, "evt_attention_stream: first frame needs v_state");
    EVT_REQUIRE(d->norm_ref == nullptr, EVT_ERR_BAD_ARG, "evt_attention_stream: norm_ref is a gated-frame output");
  } else {
    EVT_REQUIRE(d->idx && d->v_delta_t && d->v_old_t && d->kcap >= 0, EVT_ERR_BAD_ARG, "evt_attention_stream: gated frame needs idx, v_delta_t, v_old_t");
    EVT_REQUIRE(d->out_f32 != nullptr || d->store != EVT_F32, EVT_ERR_BAD_ARG,
                "evt_attention_stream: out_f32 may only be omitted with a 16-bit store type (the output then IS the pv state)");
  }
  EVT_REQUIRE(!d->qk_split || d->k_split != nullptr, EVT_ERR_BAD_ARG,
              "evt_attention_stream: qk_split needs the k_split workspace (B * H * evt_attention_stream_key_blocks * 4096 bytes)");
  if (d->B == 0) return EVT_OK;
  const bool rel = d->rel_terms != nullptr;
  {
    const int64_t need = evt_attention_stream_lds_bytes(d->store, rel ? d->gh : 0, rel ? d->gw : 0);
    EVT_REQUIRE(need > 0 && need <= EVT_LDS_PER_CU, EVT_ERR_BAD_SHAPE,
                "evt_attention_stream: a 32-row tile with a %dx%d rel-pos key grid needs %lld bytes of LDS (CU: %d); use evt_qk + "
                "evt_softmax_av_gated for this shape", d->gh, d->gw, (long long)need, EVT_LDS_PER_CU);
  }
  StreamArgs a{d->qkv, d->rel_terms, d->a_state_t, d->idx, d->count, d->v_delta_t, d->v_old_t, d->v_state, d->pv, d->out_f32,
               d->norm_ref, d->norm_parts, d->k_split, d->B, d->H, d->N, d->D, d->kcap, rel ? d->gh : 0, rel ? d->gw : 0, d->scale, d->k_split_ready,
               d->kv ? d->kv : d->qkv, Nk, d->kv ? 2 * d->D : 3 * d->D, d->kv ? 0 : d->D};

  hipStream_t s = evt_stream(stream);
  EVT_DISPATCH_STORE(d->store, T, {
    if (d->first) launch_stream<T, true>(a, d->qk_split, s);
    else launch_stream<T, false>(a, d->qk_split, s);
  });
  return evt_check_launch("evt_attention_stream");
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof_stream(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_stream_buf), sizeof(unsigned long long) * 16);
}
#endif
