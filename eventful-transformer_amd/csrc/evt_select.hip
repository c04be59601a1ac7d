// evt_select.hip -- K1: token selection from delta norms (top-k / threshold policies).
//
// One 256-thread workgroup per clip.  The clip's norms are staged once in LDS as order-preserving
// uint keys; the k-th largest key is found by a 4-pass 8-bit radix select (LDS histograms, 16 private copies per bin
// so that the norms of a clip -- which share their exponent byte -- do not serialise on one LDS atomic).  The
// selected tokens are then emitted in ASCENDING index order by a single-pass ordered compaction: every thread owns
// a contiguous chunk of tokens, counts its keys above / equal to the k-th key, an exclusive scan over the workgroup
// (`__shfl_up` wave scan + four wave totals through LDS, one barrier) gives the counts in front of the chunk, and the
// thread writes its selected tokens at their final positions.  (This replaced the round-1 wavefront-ballot
// compaction -- ballot + popcount per 256-token round, three barriers per round, 21 of them at N = 1764; the
// north_star names ballot compaction, the chunk scan does the same job with one barrier.)  Ties at the k-th key are
// resolved to the lowest token index.  The variable count of the threshold policy stays on the device.
#include "evt_common.h"

namespace {

constexpr int SEL_THREADS = 256;
constexpr int SEL_MAX_N = 16384;
constexpr int SEL_COPIES = 16;   // private histogram copies: at most 4 lanes of a wave share an LDS atomic address

// Non-negative floats order like their bit patterns.  NaN norms (bits > +inf) sort first, which
// matches ATen's topk treating NaN as the largest value.
__device__ __forceinline__ uint32_t norm_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// mode 0: top-k (k given); mode 1: threshold (norm > thr).
// parts > 0: `norms` holds `parts` partial sums of SQUARES per token (written per attention head by the fused attention
// kernel's epilogue); the norm is sqrt of their sum, added in index order (deterministic).
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const float* __restrict__ norms, int N, int k, float thr,
                                                             int mode, int kcap, int32_t* __restrict__ idx,
                                                             int32_t* __restrict__ count, int32_t* __restrict__ rest, int parts) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* keys = smem;               // N
  uint32_t* hist = smem + N;           // 256 bin totals
  uint32_t* hpriv = hist + 256;        // SEL_COPIES x 256: copy (lane & 15) of every bin
  uint32_t* wsum = hpriv + SEL_COPIES * 256;  // 8: [0..3] eq counts per wave, [4..7] selected counts per wave
  uint32_t* bc = wsum + 8;             // 4 broadcast words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  if (parts > 0) {
    const float* sq = norms + (int64_t)b * N * parts;
    for (int i = tid; i < N; i += SEL_THREADS) {
      float s = 0.f;
      for (int p = 0; p < parts; ++p) s += sq[(int64_t)i * parts + p];
      keys[i] = norm_key(sqrtf(s));
    }
  } else {
    const float* nrm = norms + (int64_t)b * N;
    for (int i = tid; i < N; i += SEL_THREADS) keys[i] = norm_key(nrm[i]);
  }
  __syncthreads();   // the compaction reads contiguous chunks: keys staged by other threads

  uint32_t kth = 0;     // key of the k-th largest element
  uint32_t need_eq = 0; // how many elements equal to kth are selected
  if (mode == 0) {
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;
#pragma unroll
    for (int c = 0; c < SEL_COPIES; ++c) hpriv[c * 256 + tid] = 0;
    __syncthreads();
    // three barriers per pass: the private copies are re-zeroed by the thread that sums them, and the two broadcast
    // words alternate between two slots, so a pass needs no barrier before the next one starts
    for (int shift = 24; shift >= 0; shift -= 8) {
      uint32_t* bcp = bc + ((shift >> 3) & 1) * 2;
      for (int i = tid; i < N; i += SEL_THREADS) {
        const uint32_t key = keys[i];
        if ((key & mask) == prefix) atomicAdd(&hpriv[(lane & (SEL_COPIES - 1)) * 256 + ((key >> shift) & 255u)], 1u);
      }
      __syncthreads();
      {
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
  // equal to the k-th key, an exclusive scan over the 256 threads (wave scan + four wave totals through LDS: one
  // barrier) gives the counts before its chunk, and it then emits its selected tokens in order:
  //   #selected before token i = #greater before i + min(#equal before i, need_eq)      (ties: lowest index first).
  // (The round-per-256-tokens version this replaces cost three barriers per round: 21 of them at N = 1764.)
  const int C = (N + SEL_THREADS - 1) / SEL_THREADS;
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
  uint32_t before = incl - mine;
  for (int w = 0; w < wave; ++w) before += wsum[w];
  const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  uint32_t gt_run = before & 0xffffu, eq_run = before >> 16;
  int32_t* out = idx + (int64_t)b * kcap;
  for (int i = i_lo; i < i_hi; ++i) {
    bool gt, eq;
    classify(keys[i], gt, eq);
    const uint32_t pos = gt_run + (eq_run < need_eq ? eq_run : need_eq);   // selected tokens before i
    const bool is_sel = gt || (eq && eq_run < need_eq);
    if (is_sel && pos < (uint32_t)kcap) out[pos] = i;
    // complement list, ascending too: #unselected before token i = i - #selected before i
    if (rest != nullptr && !is_sel) rest[(int64_t)b * N + (i - (int)pos)] = i;
    gt_run += gt;
    eq_run += eq;
  }
  const uint32_t tot_eq = total >> 16;
  const uint32_t out_run = (total & 0xffffu) + (tot_eq < need_eq ? tot_eq : need_eq);
  if (count != nullptr && tid == 0) count[b] = (int32_t)out_run;
}

int launch_select(const float* norms, int B, int N, int k, float thr, int mode, int kcap, int32_t* idx, int32_t* count,
                  int32_t* rest, void* stream, int parts = 0) {
  const size_t lds = (size_t)(N + 256 + SEL_COPIES * 256 + 8 + 4) * sizeof(uint32_t);
  EVT_ALLOW_LDS(select_kernel, lds);   // N near SEL_MAX_N: above the 64 KB default dynamic-LDS limit (set once per device)
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_THREADS), lds, evt_stream(stream), norms, N, k, thr, mode, kcap,
                     idx, count, rest, parts);
  return evt_check_launch("evt_select");
}

}  // namespace

extern "C" int evt_select_topk(const float* norms, int B, int N, int k, int32_t* idx, int32_t* rest, void* stream) {
  EVT_REQUIRE(norms != nullptr && idx != nullptr, EVT_ERR_BAD_ARG, "evt_select_topk: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0, EVT_ERR_BAD_ARG, "evt_select_topk: B=%d N=%d", B, N);
  EVT_REQUIRE(k >= 0 && k <= N, EVT_ERR_BAD_ARG, "evt_select_topk: k=%d out of range for N=%d (topk would raise)", k, N);
  EVT_REQUIRE(N <= SEL_MAX_N, EVT_ERR_BAD_SHAPE, "evt_select_topk: N=%d exceeds %d", N, SEL_MAX_N);
  if (B == 0) return EVT_OK;
  EVT_REQUIRE(k > 0 || rest == nullptr, EVT_ERR_BAD_ARG, "evt_select_topk: k == 0 with a complement list");
  if (k == 0) return EVT_OK;
  return launch_select(norms, B, N, k, 0.f, 0, k, idx, nullptr, rest, stream);
}

extern "C" int evt_select_threshold(const float* norms, int B, int N, float threshold, int kcap, int32_t* idx,
                                    int32_t* count, int32_t* rest, void* stream) {
  EVT_REQUIRE(norms != nullptr && idx != nullptr && count != nullptr, EVT_ERR_BAD_ARG, "evt_select_threshold: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0, EVT_ERR_BAD_ARG, "evt_select_threshold: B=%d N=%d", B, N);
  EVT_REQUIRE(kcap >= N, EVT_ERR_BAD_ARG, "evt_select_threshold: kcap=%d must be >= N=%d", kcap, N);
  EVT_REQUIRE(N <= SEL_MAX_N, EVT_ERR_BAD_SHAPE, "evt_select_threshold: N=%d exceeds %d", N, SEL_MAX_N);
  EVT_REQUIRE(threshold == threshold, EVT_ERR_BAD_ARG, "evt_select_threshold: NaN threshold");
  if (B == 0) return EVT_OK;
  return launch_select(norms, B, N, 0, threshold, 1, kcap, idx, count, rest, stream);
}

extern "C" int evt_select_topk_sq(const float* sq_parts, int parts, int B, int N, int k, int32_t* idx, int32_t* rest, void* stream) {
  EVT_REQUIRE(sq_parts != nullptr && idx != nullptr && parts > 0, EVT_ERR_BAD_ARG, "evt_select_topk_sq: null pointer / parts");
  EVT_REQUIRE(B >= 0 && N > 0, EVT_ERR_BAD_ARG, "evt_select_topk_sq: B=%d N=%d", B, N);
  EVT_REQUIRE(k >= 0 && k <= N, EVT_ERR_BAD_ARG, "evt_select_topk_sq: k=%d out of range for N=%d (topk would raise)", k, N);
  EVT_REQUIRE(N <= SEL_MAX_N, EVT_ERR_BAD_SHAPE, "evt_select_topk_sq: N=%d exceeds %d", N, SEL_MAX_N);
  if (B == 0) return EVT_OK;
  EVT_REQUIRE(k > 0 || rest == nullptr, EVT_ERR_BAD_ARG, "evt_select_topk_sq: k == 0 with a complement list");
  if (k == 0) return EVT_OK;
  return launch_select(sq_parts, B, N, k, 0.f, 0, k, idx, nullptr, rest, stream, parts);
}

extern "C" int evt_select_threshold_sq(const float* sq_parts, int parts, int B, int N, float threshold, int kcap, int32_t* idx,
                                       int32_t* count, int32_t* rest, void* stream) {
  EVT_REQUIRE(sq_parts != nullptr && idx != nullptr && count != nullptr && parts > 0, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: null pointer / parts");
  EVT_REQUIRE(B >= 0 && N > 0, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: B=%d N=%d", B, N);
  EVT_REQUIRE(kcap >= N, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: kcap=%d must be >= N=%d", kcap, N);
  EVT_REQUIRE(N <= SEL_MAX_N, EVT_ERR_BAD_SHAPE, "evt_select_threshold_sq: N=%d exceeds %d", N, SEL_MAX_N);
  EVT_REQUIRE(threshold == threshold, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: NaN threshold");
  if (B == 0) return EVT_OK;
  return launch_select(sq_parts, B, N, 0, threshold, 1, kcap, idx, count, rest, stream, parts);
}
