// evt_select_dev.h -- the block-wide token selection (K1) as a device function: used by select_kernel (one workgroup per
// clip, evt_select.hip) and, embedded, by the small-row-count gated linear (evt_linear_small.hip), whose workgroups each
// run the selection of their clip themselves instead of waiting for a separate launch.
#pragma once
#include "evt_common.h"

namespace {

constexpr int SEL_MAX_N = 16384;
constexpr int SEL_COPIES = 16;   // private histogram copies: at most 4 lanes of a wave share an LDS atomic address

// LDS words evt_select_block needs for N tokens
__host__ __device__ inline int evt_select_smem_words(int N) {
  const int slow = N + 256 + SEL_COPIES * 256 + 32 + 4, fast = 3 * 8 * 260 + 32;   // (fast path: 3 sets x SEL_FCOPIES x SEL_FPITCH)
  return slow > fast ? slow : fast;
}

// inclusive prefix sum over the 64 lanes of a wave on the DPP network (no LDS traffic, unlike __shfl_up)
__device__ __forceinline__ uint32_t evt_wave_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);   // row_shr:4, lanes 4..15 of each row
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);   // row_shr:8, lanes 8..15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return v;
}

// Non-negative floats order like their bit patterns.  NaN norms (bits > +inf) sort first, which
// matches ATen's topk treating NaN as the largest value.
__device__ __forceinline__ uint32_t norm_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// Selects from the N norms of ONE clip.  mode 0: top-k (k given); mode 1: threshold (norm > thr).
// parts > 0: `norms` holds `parts` partial sums of SQUARES per token (written per attention head by the fused attention
// kernel's epilogue); the norm is sqrt of their sum, added in index order (deterministic).
// smem: evt_select_smem_words(N) words of LDS, free on entry (the function synchronises before touching it only through
// its own barriers: the caller must have finished with the region).  Outputs (any address space; nullable except out_idx):
// out_idx[0 .. count) ascending selected tokens (at most kcap), out_idx2 a second copy, out_rest the complement list,
// out_count the count.  Returns the count to every thread.  All THREADS threads of the workgroup must call it.
template <int THREADS>
__device__ __forceinline__ int evt_select_block(const float* __restrict__ norms, int parts, int N, int k, float thr, int mode,
                                                int kcap, uint32_t* smem, int32_t* out_idx, int32_t* out_idx2,
                                                int32_t* out_count, int32_t* out_rest) {
  constexpr int W = THREADS / 64;
  static_assert(THREADS >= 256 && THREADS % 64 == 0 && W <= 16, "256 .. 1024 threads");
  constexpr int CMAX = 8;   // fast path: every thread keeps its contiguous chunk of <= 8 keys in registers
  if (N <= CMAX * THREADS) {
    // ---- register-resident selection (N <= 8 x THREADS: every per-stream shape) -------------------------------------------
    // Thread t owns the contiguous tokens [t C, (t + 1) C) for the radix passes AND the ordered compaction, so the keys never
    // go through LDS.  A radix pass is: LDS atomics into one of THREE histogram sets ([bin][16 copies]), ONE barrier, then
    // every wave on its own sums the copies (16-byte reads), suffix-scans the 256 bins on the DPP network (lane l owns the
    // bins 252 - 4 l .. 255 - 4 l) and finds the bin holding the k-th key by ballot + readlane -- no second and third
    // barrier, no ds_bpermute shuffles.  The set that was read one pass ago is zeroed meanwhile (nobody touches it until
    // the pass after next).  6 barriers per selection instead of 15; ~9 -> ~3 us inside a launch.
    // histogram set: SEL_FCOPIES private copies (copy = lane & 7) of 256 bins, copy pitch 260 words: the 8 copies of a bin sit
    // in 8 different banks (an atomic's lanes spread), and a lane's 4 consecutive bins of one copy are one aligned 16-byte read,
    // contiguous over the lanes (conflict-free; [bin][copy] made every read a 16-way bank conflict in all 8 waves at once)
    constexpr int SEL_FCOPIES = 8, SEL_FPITCH = 260, SET = SEL_FCOPIES * SEL_FPITCH;
    uint32_t* hs = smem;                        // 3 sets
    uint32_t* wsum2 = smem + 3 * SET;           // W scan partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = (N + THREADS - 1) / THREADS;
    const int i_lo = tid * C;
    uint32_t key[CMAX];
    if (parts == 12 && (reinterpret_cast<uintptr_t>(norms) & 15) == 0) {
      // ViT-B's 12 heads: a token's partials are 48 consecutive bytes.  All 3 x C 16-byte loads of the thread are requested
      // together through clamped addresses (a scalar load per partial inside the predicated `i < N` block was one round trip
      // per partial: 16.1 vs 6.8 us per launch at N = 1764, 23.6 vs 3.9 us at N = 4096), then summed in index order as before.
      float4 x[CMAX][3];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        const int i = i_lo + c;
        const float4* pp = reinterpret_cast<const float4*>(norms + (int64_t)((c < C && i < N) ? i : 0) * 12);
#pragma unroll
        for (int q = 0; q < 3; ++q) x[c][q] = pp[q];
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) { sq += x[c][q].x; sq += x[c][q].y; sq += x[c][q].z; sq += x[c][q].w; }
        key[c] = (c < C && i_lo + c < N) ? norm_key(sqrtf(sq)) : 0u;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        const int i = i_lo + c;
        key[c] = 0;
        if (c < C && i < N) {
          if (parts > 0) {
            float sq = 0.f;
            for (int p = 0; p < parts; ++p) sq += norms[(int64_t)i * parts + p];
            key[c] = norm_key(sqrtf(sq));
          } else {
            key[c] = norm_key(norms[i]);
          }
        }
      }
    }
    uint32_t kth = 0, need_eq = 0, kmask = 0xffffffffu;   // keys are compared on the digits the passes have resolved
    if (mode == 0) {
      for (int e = tid; e < 3 * SET; e += THREADS) hs[e] = 0;
      __syncthreads();
      uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;
      int pass = 0;
      for (int shift = 24; shift >= 0; shift -= 8, ++pass) {
        uint32_t* set = hs + (pass % 3) * SET;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C && i_lo + c < N && (key[c] & mask) == prefix)
            atomicAdd(&set[(lane & (SEL_FCOPIES - 1)) * SEL_FPITCH + ((key[c] >> shift) & 255u)], 1u);
        __syncthreads();
        {   // zero the set used by the pass before this one (its readers all passed the barrier above)
          uint32_t* zs = hs + ((pass + 2) % 3) * SET;
          if (pass > 0 && pass < 3)
            for (int e = tid; e < SET; e += THREADS) zs[e] = 0;
        }
        // every wave: bins of this lane in DESCENDING order d = 0..3 <-> bin 255 - 4 lane - d
        uint32_t hb[4] = {0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < SEL_FCOPIES; ++v) {
          const uint4 x = *reinterpret_cast<const uint4*>(set + v * SEL_FPITCH + 252 - 4 * lane);   // bins 252 - 4 lane .. + 3
          hb[0] += x.w; hb[1] += x.z; hb[2] += x.y; hb[3] += x.x;
        }
        const uint32_t mine = (hb[0] + hb[1]) + (hb[2] + hb[3]);
        const uint32_t incl = evt_wave_scan_u32(mine);      // keys in the bins of lanes 0 .. lane, i.e. in all HIGHER-or-equal bins
        const uint32_t higher = incl - mine;
        const bool here = higher < remaining && remaining <= incl;   // exactly one lane
        uint32_t acc = higher, bin = 0, rank = 0, pop = 0;
        if (here) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            if (rank == 0 && acc + hb[d] >= remaining) { bin = 255u - 4u * lane - d; rank = remaining - acc; pop = hb[d]; }
            acc += hb[d];
          }
        }
        const unsigned long long bal = __ballot(here);
        const int src = __ffsll((long long)bal) - 1;
        bin = (uint32_t)__builtin_amdgcn_readlane((int)bin, src);
        rank = (uint32_t)__builtin_amdgcn_readlane((int)rank, src);
        pop = (uint32_t)__builtin_amdgcn_readlane((int)pop, src);
        prefix |= bin << shift;
        mask |= 255u << shift;
        remaining = rank;
        // EVERY key of the chosen bin is selected (with distinct norms: the bin holds one key after three passes): the lower
        // digits cannot change the selection -- "above the prefix" and "in the bin" decide, lowest index first among equals
        // as before.  Every wave takes the same decision from the same histogram.
        if (pop == rank) break;
      }
      kth = prefix;
      kmask = mask;
      need_eq = remaining;
    }
    // ordered compaction from the registers
    const float thr_f = thr;
    uint32_t my_gt = 0, my_eq = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C && i_lo + c < N) {
        const bool gt = mode == 0 ? (key[c] & kmask) > kth : __uint_as_float(key[c]) > thr_f;
        const bool eq = mode == 0 && (key[c] & kmask) == kth;
        my_gt += gt;
        my_eq += eq;
      }
    }
    const uint32_t mine = (my_eq << 16) | my_gt;   // N <= 8192 < 65536: the halves cannot carry into each other
    const uint32_t incl = evt_wave_scan_u32(mine);
    if (lane == 63) wsum2[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const uint32_t ws = wsum2[w];
      if (w < wave) before += ws;
      total += ws;
    }
    uint32_t gt_run = before & 0xffffu, eq_run = before >> 16;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const int i = i_lo + c;
      if (c < C && i < N) {
        const bool gt = mode == 0 ? (key[c] & kmask) > kth : __uint_as_float(key[c]) > thr_f;
        const bool eq = mode == 0 && (key[c] & kmask) == kth;
        const uint32_t pos = gt_run + (eq_run < need_eq ? eq_run : need_eq);   // selected tokens before i
        const bool is_sel = gt || (eq && eq_run < need_eq);
        if (is_sel && pos < (uint32_t)kcap) {
          out_idx[pos] = i;
          if (out_idx2 != nullptr) out_idx2[pos] = i;
        }
        if (out_rest != nullptr && !is_sel) out_rest[i - (int)pos] = i;
        gt_run += gt;
        eq_run += eq;
      }
    }
    const uint32_t tot_eq = total >> 16;
    const uint32_t out_run = (total & 0xffffu) + (tot_eq < need_eq ? tot_eq : need_eq);
    if (out_count != nullptr && tid == 0) *out_count = (int32_t)out_run;
    return (int)out_run;
  }
  uint32_t* keys = smem;               // N
  uint32_t* hist = smem + N;           // 256 bin totals
  uint32_t* hpriv = hist + 256;        // SEL_COPIES x 256: copy (lane & 15) of every bin
  uint32_t* wsum = hpriv + SEL_COPIES * 256;  // W scan partials
  uint32_t* bc = wsum + 32;            // 4 broadcast words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (parts > 0) {
    for (int i = tid; i < N; i += THREADS) {
      float s = 0.f;
      for (int p = 0; p < parts; ++p) s += norms[(int64_t)i * parts + p];
      keys[i] = norm_key(sqrtf(s));
    }
  } else {
    for (int i = tid; i < N; i += THREADS) keys[i] = norm_key(norms[i]);
  }
  if (mode == 0 && tid < 256) {
#pragma unroll
    for (int c = 0; c < SEL_COPIES; ++c) hpriv[c * 256 + tid] = 0;
  }
  __syncthreads();   // the compaction reads contiguous chunks: keys staged by other threads

  uint32_t kth = 0;     // key of the k-th largest element
  uint32_t need_eq = 0; // how many elements equal to kth are selected
  if (mode == 0) {
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;
    // three barriers per pass: the private copies are re-zeroed by the thread that sums them, and the two broadcast
    // words alternate between two slots, so a pass needs no barrier before the next one starts
    for (int shift = 24; shift >= 0; shift -= 8) {
      uint32_t* bcp = bc + ((shift >> 3) & 1) * 2;
      for (int i = tid; i < N; i += THREADS) {
        const uint32_t key = keys[i];
        if ((key & mask) == prefix) atomicAdd(&hpriv[(lane & (SEL_COPIES - 1)) * 256 + ((key >> shift) & 255u)], 1u);
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t tot = 0;
#pragma unroll
        for (int c = 0; c < SEL_COPIES; ++c) { tot += hpriv[c * 256 + tid]; hpriv[c * 256 + tid] = 0; }
        hist[tid] = tot;
      }
      __syncthreads();
      if (wave == 0) {
        // Each lane owns 4 consecutive bins; suffix-scan from the top bin.
        const uint32_t h0 = hist[lane * 4 + 0], h1 = hist[lane * 4 + 1], h2 = hist[lane * 4 + 2], h3 = hist[lane * 4 + 3];
        const uint32_t mine = h0 + h1 + h2 + h3;
        uint32_t above = mine;  // inclusive suffix sum over lanes >= this lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = __shfl_down(above, o, 64);
          if (lane + o < 64) above += t;
        }
        const uint32_t higher = above - mine;  // elements in bins owned by higher lanes
        // The k-th element lives in this lane's bins iff higher < remaining <= higher + mine.
        if (higher < remaining && remaining <= higher + mine) {
          uint32_t acc = higher;
          int bin = lane * 4 + 3;
          uint32_t hb[4] = {h0, h1, h2, h3};
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            if (acc + hb[j] >= remaining) { bin = lane * 4 + j; break; }
            acc += hb[j];
          }
          bcp[0] = (uint32_t)bin;
          bcp[1] = remaining - acc;  // rank inside the chosen bin (1-based)
        }
      }
      __syncthreads();
      prefix |= bcp[0] << shift;
      mask |= 255u << shift;
      remaining = bcp[1];
    }
    kth = prefix;
    need_eq = remaining;
  }

  // Ordered compaction in ONE pass: thread t owns the contiguous tokens [t*C, (t+1)*C).  It counts its keys above and
  // equal to the k-th key, an exclusive scan over the threads (wave scan + the wave totals through LDS: one
  // barrier) gives the counts before its chunk, and it then emits its selected tokens in order:
  //   #selected before token i = #greater before i + min(#equal before i, need_eq)      (ties: lowest index first).
  const int C = (N + THREADS - 1) / THREADS;
  const int i_lo = tid * C, i_hi = min(N, i_lo + C);
  const float thr_f = thr;
  auto classify = [&](uint32_t key, bool& gt, bool& eq) {
    if (mode == 0) { gt = key > kth; eq = key == kth; }
    else { gt = __uint_as_float(key) > thr_f; eq = false; }   // norm.gt(threshold), policies.py:28
  };
  uint32_t my_gt = 0, my_eq = 0;
  for (int i = i_lo; i < i_hi; ++i) {
    bool gt, eq;
    classify(keys[i], gt, eq);
    my_gt += gt;
    my_eq += eq;
  }
  // exclusive scan of (eq << 16 | gt) over the workgroup (N <= 16384 < 65536: the halves cannot carry into each other)
  const uint32_t mine = (my_eq << 16) | my_gt;
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t before = incl - mine, total = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const uint32_t ws = wsum[w];
    if (w < wave) before += ws;
    total += ws;
  }
  uint32_t gt_run = before & 0xffffu, eq_run = before >> 16;
  for (int i = i_lo; i < i_hi; ++i) {
    bool gt, eq;
    classify(keys[i], gt, eq);
    const uint32_t pos = gt_run + (eq_run < need_eq ? eq_run : need_eq);   // selected tokens before i
    const bool is_sel = gt || (eq && eq_run < need_eq);
    if (is_sel && pos < (uint32_t)kcap) {
      out_idx[pos] = i;
      if (out_idx2 != nullptr) out_idx2[pos] = i;
    }
    // complement list, ascending too: #unselected before token i = i - #selected before i
    if (out_rest != nullptr && !is_sel) out_rest[i - (int)pos] = i;
    gt_run += gt;
    eq_run += eq;
  }
  const uint32_t tot_eq = total >> 16;
  const uint32_t out_run = (total & 0xffffu) + (tot_eq < need_eq ? tot_eq : need_eq);
  if (out_count != nullptr && tid == 0) *out_count = (int32_t)out_run;
  return (int)out_run;
}

}  // namespace
