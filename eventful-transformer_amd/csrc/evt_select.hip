// evt_select.hip -- K1: token selection from delta norms (top-k / threshold policies).
//
// One 256-thread workgroup per clip.  The clip's norms are staged once in LDS as order-preserving
// uint keys; the k-th largest key is found by a 4-pass 8-bit radix select (LDS histograms); the
// selected tokens are then emitted in ASCENDING index order by wavefront-ballot compaction:
// per 256-token round each wave ranks its lanes with a 64-bit ballot + popcount of the lower
// lanes, and the 4 per-wave counts are combined through LDS.  Ties at the k-th key are resolved
// to the lowest token index.  The variable count of the threshold policy stays on the device.
#include "evt_common.h"

namespace {

constexpr int SEL_THREADS = 256;
constexpr int SEL_MAX_N = 16384;

// Non-negative floats order like their bit patterns.  NaN norms (bits > +inf) sort first, which
// matches ATen's topk treating NaN as the largest value.
__device__ __forceinline__ uint32_t norm_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// mode 0: top-k (k given); mode 1: threshold (norm > thr).
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const float* __restrict__ norms, int N, int k, float thr,
                                                             int mode, int kcap, int32_t* __restrict__ idx,
                                                             int32_t* __restrict__ count, int32_t* __restrict__ rest) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* keys = smem;               // N
  uint32_t* hist = smem + N;           // 256
  uint32_t* wsum = hist + 256;         // 8: [0..3] eq counts per wave, [4..7] selected counts per wave
  uint32_t* bc = wsum + 8;             // 4 broadcast words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* nrm = norms + (int64_t)b * N;
  for (int i = tid; i < N; i += SEL_THREADS) keys[i] = norm_key(nrm[i]);

  uint32_t kth = 0;     // key of the k-th largest element
  uint32_t need_eq = 0; // how many elements equal to kth are selected
  if (mode == 0) {
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < N; i += SEL_THREADS) {
        const uint32_t key = keys[i];
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
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
          bc[0] = (uint32_t)bin;
          bc[1] = remaining - acc;  // rank inside the chosen bin (1-based)
        }
      }
      __syncthreads();
      prefix |= bc[0] << shift;
      mask |= 255u << shift;
      remaining = bc[1];
      __syncthreads();
    }
    kth = prefix;
    need_eq = remaining;
  }

  // Ordered compaction.
  uint32_t eq_run = 0, out_run = 0;  // running totals, identical in every thread
  int32_t* out = idx + (int64_t)b * kcap;
  for (int base = 0; base < N; base += SEL_THREADS) {
    const int i = base + tid;
    bool is_eq = false, is_sel = false;
    uint32_t key = 0;
    if (i < N) {
      key = keys[i];
      if (mode == 0) {
        is_eq = (key == kth);
        is_sel = (key > kth);
      } else {
        is_sel = __uint_as_float(key) > thr;  // norm.gt(threshold), policies.py:28
      }
    }
    const unsigned long long lower = (1ull << lane) - 1ull;
    uint32_t eq_rank = 0;
    if (mode == 0) {
      const unsigned long long beq = __ballot(is_eq);
      if (lane == 0) wsum[wave] = (uint32_t)__popcll(beq);
      __syncthreads();
      eq_rank = eq_run + (uint32_t)__popcll(beq & lower);
      for (int w = 0; w < wave; ++w) eq_rank += wsum[w];
      if (is_eq && eq_rank < need_eq) is_sel = true;
      eq_run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    const unsigned long long bsel = __ballot(is_sel);
    if (lane == 0) wsum[4 + wave] = (uint32_t)__popcll(bsel);
    __syncthreads();
    uint32_t pos = out_run + (uint32_t)__popcll(bsel & lower);
    for (int w = 0; w < wave; ++w) pos += wsum[4 + w];
    if (is_sel && pos < (uint32_t)kcap) out[pos] = i;
    // complement list, ascending too: #unselected before token i = i - #selected before i
    if (rest != nullptr && i < N && !is_sel) rest[(int64_t)b * N + (i - (int)pos)] = i;
    out_run += wsum[4] + wsum[5] + wsum[6] + wsum[7];
    __syncthreads();
  }
  if (count != nullptr && tid == 0) count[b] = (int32_t)out_run;
}

int launch_select(const float* norms, int B, int N, int k, float thr, int mode, int kcap, int32_t* idx, int32_t* count,
                  int32_t* rest, void* stream) {
  const size_t lds = (size_t)(N + 256 + 8 + 4) * sizeof(uint32_t);
  if (lds > 64 * 1024)  // N near SEL_MAX_N: above the 64 KB default dynamic-LDS limit
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_THREADS), lds, evt_stream(stream), norms, N, k, thr, mode, kcap,
                     idx, count, rest);
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
