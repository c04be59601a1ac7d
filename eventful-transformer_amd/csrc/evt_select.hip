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
#include "evt_select_dev.h"

namespace {

// THREADS: 256 up to 2048 tokens, 1024 above (<= 8 tokens per thread: the register-resident path of evt_select_block up to
// 8192 tokens; N = 4096 top-k at 256 threads took the LDS-key path: 13.3 us vs 6.8 us for N = 1764)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void select_kernel(const float* __restrict__ norms, int N, int k, float thr,
                                                         int mode, int kcap, int32_t* __restrict__ idx,
                                                         int32_t* __restrict__ count, int32_t* __restrict__ rest, int parts) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int b = blockIdx.x;
  evt_select_block<THREADS>(norms + (int64_t)b * N * (parts > 0 ? parts : 1), parts, N, k, thr, mode, kcap, smem,
                            idx + (int64_t)b * kcap, nullptr, count ? count + b : nullptr, rest ? rest + (int64_t)b * N : nullptr);
}

int launch_select(const float* norms, int B, int N, int k, float thr, int mode, int kcap, int32_t* idx, int32_t* count,
                  int32_t* rest, void* stream, int parts = 0) {
  const size_t lds = (size_t)evt_select_smem_words(N) * sizeof(uint32_t);
  if (N <= 2048) {
    EVT_ALLOW_LDS(select_kernel<256>, lds);   // N near SEL_MAX_N: above the 64 KB default dynamic-LDS limit (set once per device)
    hipLaunchKernelGGL(select_kernel<256>, dim3(B), dim3(256), lds, evt_stream(stream), norms, N, k, thr, mode, kcap, idx, count, rest, parts);
  } else {
    EVT_ALLOW_LDS(select_kernel<1024>, lds);
    hipLaunchKernelGGL(select_kernel<1024>, dim3(B), dim3(1024), lds, evt_stream(stream), norms, N, k, thr, mode, kcap, idx, count, rest, parts);
  }
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
