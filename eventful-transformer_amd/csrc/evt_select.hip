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
#include <algorithm>
#include <stdlib.h>

namespace {

// THREADS: 256 up to 2048 tokens, 1024 above (<= 8 tokens per thread: the register-resident path of evt_select_block up to
// 8192 tokens; N = 4096 top-k at 256 threads took the LDS-key path: 13.3 us vs 6.8 us for N = 1764)
// Prefetch rider (evt_select_prefetch_next): workgroups [B, gridDim.x) of the launch read a read-only operand -- the weight planes
// of a gated linear a few launches ahead -- so that it finds them in the memory-side cache.  The selection of one video stream
// is ONE workgroup on a 256-CU chip for ~5 us: the riders use the rest of it and the memory system nobody else is using, and
// the launch does not get longer.  (A stand-alone prefetch launch costs its ~4.7 us back; on a side stream of the frame's HIP
// graph the fork / join at every block cost 0.4 ms per frame.)
struct SelRider { const uint4* p[2]; int64_t n16[2]; uint32_t* sink; uint32_t magic; int deep; };   // up to two ranges

template <int THREADS>
__global__ __launch_bounds__(THREADS) void select_kernel(const float* __restrict__ norms, int N, int k, float thr,
                                                         int mode, int kcap, int32_t* __restrict__ idx,
                                                         int32_t* __restrict__ count, int32_t* __restrict__ rest, int parts,
                                                         int B, SelRider rider) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  if ((int)blockIdx.x >= B) {
    uint32_t acc = 0;
    const int64_t stride = (int64_t)(gridDim.x - B) * THREADS;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint4* __restrict__ pp = rider.p[q];
      const int64_t n16 = rider.n16[q];
      int64_t i = (int64_t)(blockIdx.x - B) * THREADS + threadIdx.x;
      // EIGHT independent 16-byte loads in flight per thread, 65536 threads: ~8 MB on the way -- at ~2 us of cold-read latency
      // the planes of one gated linear (7-9 MB) arrive within the selection's own ~5 us (4 loads x 32768 threads took ~9 us)
      if (rider.deep)
      for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
      }
      for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = pp[i], b = pp[i + stride], c = pp[i + 2 * stride], d = pp[i + 3 * stride];
        acc ^= (a.x ^ a.y ^ a.z ^ a.w) ^ (b.x ^ b.y ^ b.z ^ b.w) ^ (c.x ^ c.y ^ c.z ^ c.w) ^ (d.x ^ d.y ^ d.z ^ d.w);
      }
      for (; i < n16; i += stride) { const uint4 a = pp[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    }
    if (acc == rider.magic) *rider.sink = acc;   // (a run-time word nobody's data xors to: the loads must not be optimised away)
    return;
  }
  const int b = blockIdx.x;
  evt_select_block<THREADS>(norms + (int64_t)b * N * (parts > 0 ? parts : 1), parts, N, k, thr, mode, kcap, smem,
                            idx + (int64_t)b * kcap, nullptr, count ? count + b : nullptr, rest ? rest + (int64_t)b * N : nullptr);
}

thread_local SelRider g_rider = {{nullptr, nullptr}, {0, 0}, nullptr, 0, 0};   // armed by evt_select_prefetch_next, consumed by the next launch

// a selection that launches nothing must not leave its rider armed for some later launch (the range may be gone by then)
void drop_rider() { g_rider = {{nullptr, nullptr}, {0, 0}, nullptr, 0, 0}; }
// Every evt_select_* entry point holds one of these: whichever way it returns (a launch, an early return, a failed
// EVT_REQUIRE), the rider armed for it is gone afterwards -- a stale pointer must never ride on some later, unrelated launch.
struct RiderScope { ~RiderScope() { drop_rider(); } };

int launch_select(const float* norms, int B, int N, int k, float thr, int mode, int kcap, int32_t* idx, int32_t* count,
                  int32_t* rest, void* stream, int parts = 0) {
  const size_t lds = (size_t)evt_select_smem_words(N) * sizeof(uint32_t);
  const SelRider rider = g_rider;
  g_rider = {{nullptr, nullptr}, {0, 0}, nullptr, 0, 0};
  const int threads = N <= 2048 ? 256 : 1024;
  const int64_t n16 = rider.n16[0] + rider.n16[1];
  // Rider workgroups: at most 128 -- half the CUs, so that none of them shares a CU with the selection's own workgroup(s), which
  // set the launch's length (672^2 frame: 1.62 ms with 128 riders of 256 threads, 1.67 with 256; 1.69 without riders).
  constexpr int rider_wgs = 128;
  const int per = rider.deep ? 8 : 4;
  const int extra = n16 == 0 ? 0 : (int)std::min<int64_t>((n16 + per * threads - 1) / (per * threads), std::max(1, rider_wgs));
  if (N <= 2048) {
    EVT_ALLOW_LDS(select_kernel<256>, lds);   // N near SEL_MAX_N: above the 64 KB default dynamic-LDS limit (set once per device)
    hipLaunchKernelGGL(select_kernel<256>, dim3(B + extra), dim3(256), lds, evt_stream(stream), norms, N, k, thr, mode, kcap, idx, count, rest, parts, B, rider);
  } else {
    EVT_ALLOW_LDS(select_kernel<1024>, lds);
    hipLaunchKernelGGL(select_kernel<1024>, dim3(B + extra), dim3(1024), lds, evt_stream(stream), norms, N, k, thr, mode, kcap, idx, count, rest, parts, B, rider);
  }
  return evt_check_launch("evt_select");
}

}  // namespace

extern "C" int evt_select_prefetch_next(const void* ptr, int64_t bytes, void* sink) {
  EVT_REQUIRE(ptr != nullptr && sink != nullptr, EVT_ERR_BAD_ARG, "evt_select_prefetch_next: null pointer");
  EVT_REQUIRE(bytes >= 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0, EVT_ERR_BAD_ARG,
              "evt_select_prefetch_next: bytes=%lld / pointer must be 16-byte aligned", (long long)bytes);
  const int q = g_rider.n16[0] == 0 ? 0 : 1;   // a second call before the launch arms a second range (a third replaces it)
  g_rider.p[q] = reinterpret_cast<const uint4*>(ptr);
  g_rider.n16[q] = bytes / 16;
  g_rider.sink = reinterpret_cast<uint32_t*>(sink);
  g_rider.magic = 0x9e3779b9u;
  g_rider.deep = 1;   // 8 x 16-byte loads in flight per thread (4: measured slower, profiles/r04/prefetch_rider_shapes.txt)
  return EVT_OK;
}

extern "C" int evt_select_topk(const float* norms, int B, int N, int k, int32_t* idx, int32_t* rest, void* stream) {
  RiderScope rider_scope;
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
  RiderScope rider_scope;
  EVT_REQUIRE(norms != nullptr && idx != nullptr && count != nullptr, EVT_ERR_BAD_ARG, "evt_select_threshold: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0, EVT_ERR_BAD_ARG, "evt_select_threshold: B=%d N=%d", B, N);
  EVT_REQUIRE(kcap >= N, EVT_ERR_BAD_ARG, "evt_select_threshold: kcap=%d must be >= N=%d", kcap, N);
  EVT_REQUIRE(N <= SEL_MAX_N, EVT_ERR_BAD_SHAPE, "evt_select_threshold: N=%d exceeds %d", N, SEL_MAX_N);
  EVT_REQUIRE(threshold == threshold, EVT_ERR_BAD_ARG, "evt_select_threshold: NaN threshold");
  if (B == 0) return EVT_OK;
  return launch_select(norms, B, N, 0, threshold, 1, kcap, idx, count, rest, stream);
}

extern "C" int evt_select_topk_sq(const float* sq_parts, int parts, int B, int N, int k, int32_t* idx, int32_t* rest, void* stream) {
  RiderScope rider_scope;
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
  RiderScope rider_scope;
  EVT_REQUIRE(sq_parts != nullptr && idx != nullptr && count != nullptr && parts > 0, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: null pointer / parts");
  EVT_REQUIRE(B >= 0 && N > 0, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: B=%d N=%d", B, N);
  EVT_REQUIRE(kcap >= N, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: kcap=%d must be >= N=%d", kcap, N);
  EVT_REQUIRE(N <= SEL_MAX_N, EVT_ERR_BAD_SHAPE, "evt_select_threshold_sq: N=%d exceeds %d", N, SEL_MAX_N);
  EVT_REQUIRE(threshold == threshold, EVT_ERR_BAD_ARG, "evt_select_threshold_sq: NaN threshold");
  if (B == 0) return EVT_OK;
  return launch_select(sq_parts, B, N, 0, threshold, 1, kcap, idx, count, rest, stream, parts);
}
