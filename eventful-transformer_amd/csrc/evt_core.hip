// evt_core.hip -- version / error plumbing of libevt_hip.so.
#include "evt_common.h"
#include <algorithm>

static thread_local char g_err[512] = "";

char* evt_err_buf() { return g_err; }

int evt_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static thread_local char g_note[256] = "";

void evt_note_launch_problem(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_note, sizeof(g_note), fmt, ap);
  va_end(ap);
}

int evt_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char note[256];
    snprintf(note, sizeof(note), "%s", g_note);
    g_note[0] = 0;
    return note[0] ? evt_fail(EVT_ERR_HIP, "%s: %s (%s)", what, hipGetErrorString(e), note)
                   : evt_fail(EVT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  }
  g_note[0] = 0;
  return EVT_OK;
}

static thread_local int g_cu_budget = 0;
int evt_cu_budget() { return g_cu_budget; }
extern "C" int evt_set_cu_budget(int32_t cus) {
  if (cus < 0) return evt_fail(EVT_ERR_BAD_ARG, "evt_set_cu_budget: cus=%d", (int)cus);
  g_cu_budget = cus;
  return EVT_OK;
}

extern "C" int evt_version(void) { return EVT_ABI_VERSION; }
extern "C" const char* evt_last_error_string(void) { return g_err; }
extern "C" const char* evt_target_arch(void) { return "gfx950"; }

// evt_prefetch: read `bytes` of a read-only operand (a layer's weight planes) so that the launch that streams them a few
// microseconds later finds them in the memory-side cache instead of HBM.  One video stream runs 48 gated linears per frame over
// 340 MB of weight planes -- more than any cache holds from one frame to the next -- and each of them is a 6-14 us launch whose
// workgroups wait out cold-read latency with ~128 KB in flight per CU: graph-replayed with cold planes they take 0.7-3.8 us
// longer than with hot ones (scripts/probes/gemm_cold_weights.py).  The caller issues this on a side stream while the
// previous block's launches (single-workgroup selections, latency-bound attention) leave the memory system idle.
namespace {
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* __restrict__ p, int64_t n16, uint32_t* __restrict__ sink, uint32_t magic) {
  uint32_t acc = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {   // four independent 16-byte loads in flight per thread
    const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc ^= (a.x ^ a.y ^ a.z ^ a.w) ^ (b.x ^ b.y ^ b.z ^ b.w) ^ (c.x ^ c.y ^ c.z ^ c.w) ^ (d.x ^ d.y ^ d.z ^ d.w);
  }
  for (; i < n16; i += stride) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == magic) *sink = acc;   // (a run-time word nobody's data xors to: the loads must not be optimised away)
}
}  // namespace

extern "C" int evt_prefetch(const void* ptr, int64_t bytes, void* sink, void* stream) {
  EVT_REQUIRE(ptr != nullptr && sink != nullptr, EVT_ERR_BAD_ARG, "evt_prefetch: null pointer");
  EVT_REQUIRE(bytes >= 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0, EVT_ERR_BAD_ARG, "evt_prefetch: bytes=%lld / pointer must be 16-byte aligned", (long long)bytes);
  const int64_t n16 = bytes / 16;
  if (n16 == 0) return EVT_OK;
  const int64_t want = (n16 + 4 * 256 - 1) / (4 * 256);
  const unsigned grid = (unsigned)std::min<int64_t>(want, 128);   // a fraction of the chip: it runs beside another stream's launches
  hipLaunchKernelGGL(prefetch_kernel, dim3(grid), dim3(256), 0, evt_stream(stream), reinterpret_cast<const uint4*>(ptr), n16, reinterpret_cast<uint32_t*>(sink), 0x9e3779b9u);
  return evt_check_launch("evt_prefetch");
}
