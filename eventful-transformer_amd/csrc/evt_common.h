// evt_common.h -- shared device/host helpers for libevt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/evt_abi.h"

#define EVT_WAVE 64

// ---------------------------------------------------------------------------------------------
// host side: error reporting (thread-local, no allocation)
// ---------------------------------------------------------------------------------------------
char* evt_err_buf();
int evt_fail(int code, const char* fmt, ...);
int evt_check_launch(const char* what);
// A host-side preparation step of the launch that follows failed (e.g. raising a kernel's dynamic-LDS limit): the note is
// thread-local and evt_check_launch reports it together with the launch error it explains.
void evt_note_launch_problem(const char* fmt, ...);

#define EVT_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return evt_fail((code), __VA_ARGS__); \
  } while (0)

static inline hipStream_t evt_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Per-DEVICE host caches (one process may drive several GPUs; hipFuncSetAttribute applies to the current device only).
constexpr int EVT_MAX_DEVICES = 64;
static inline int evt_current_device() {   // -1: outside the per-device tables (never aliased onto another device's slot)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  return dev >= 0 && dev < EVT_MAX_DEVICES ? dev : -1;
}
// Compute units a persistent launch sizes its grid for: the calling thread's budget (evt_set_cu_budget: the launch goes to a
// stream with a CU mask), else the CUs of the current device (cached per device; 256 if the runtime does not say).
int evt_cu_budget();   // evt_core.hip: thread-local, 0 = none
static inline int evt_cu_count() {
  static std::atomic<int> cus[EVT_MAX_DEVICES];
  const int budget = evt_cu_budget();
  if (budget > 0) return budget;
  int dev = evt_current_device();
  if (dev >= 0 && cus[dev].load(std::memory_order_relaxed) > 0) return cus[dev].load(std::memory_order_relaxed);
  int real = 0, n = 0;
  if (hipGetDevice(&real) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, real) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0) cus[dev].store(n, std::memory_order_relaxed);
  return n;
}
constexpr int EVT_LDS_PER_CU = 160 * 1024;   // gfx950

// Raises a kernel's dynamic-LDS limit to `bytes` when that exceeds what this call site has already set on the current
// device (one static table per call site = per kernel instantiation): no runtime call on the launch path afterwards.
// The table only remembers a size the runtime ACCEPTED (a refused raise is retried by the next call and its reason is kept for
// evt_check_launch); entries are atomics, so host threads driving the library concurrently race at worst to the same call.
// Devices beyond the table are not cached: every launch there makes the runtime call.
#define EVT_ALLOW_LDS(kernel, bytes)                                                                            \
  do {                                                                                                           \
    static std::atomic<int> evt_lds_set_[EVT_MAX_DEVICES];                                                       \
    const int evt_dev_ = evt_current_device();                                                                   \
    const int evt_want_ = (int)(bytes);                                                                          \
    if (evt_want_ > 64 * 1024 && (evt_dev_ < 0 || evt_lds_set_[evt_dev_].load(std::memory_order_relaxed) < evt_want_)) { \
      hipError_t evt_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, evt_want_); \
      if (evt_e_ != hipSuccess) {                                                                                \
        (void)hipGetLastError();                                                                                 \
        evt_note_launch_problem("hipFuncSetAttribute(%s, dynamic LDS %d bytes) failed: %s", #kernel, evt_want_, hipGetErrorString(evt_e_)); \
      } else if (evt_dev_ >= 0) {                                                                                \
        int evt_old_ = evt_lds_set_[evt_dev_].load(std::memory_order_relaxed);                                   \
        while (evt_old_ < evt_want_ && !evt_lds_set_[evt_dev_].compare_exchange_weak(evt_old_, evt_want_)) {}    \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// DPP reductions (no LDS traffic, unlike ds_bpermute-based __shfl_xor): row_shr 1/2/4/8 leave the total of each
// 16-lane row in its lane 15; row_bcast15 / row_bcast31 fold the four rows into lane 63.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float evt_dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
// total of each 16-lane row, valid in the row's lane 15
__device__ __forceinline__ float row16_sum_dpp(float v) {
  v += evt_dpp<0x111, 0xf, 0xf>(0.f, v);
  v += evt_dpp<0x112, 0xf, 0xf>(0.f, v);
  v += evt_dpp<0x114, 0xf, 0xe>(0.f, v);
  v += evt_dpp<0x118, 0xf, 0xc>(0.f, v);
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum_dpp(v);
  v += evt_dpp<0x142, 0xa, 0xf>(0.f, v);
  v += evt_dpp<0x143, 0xc, 0xf>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, evt_dpp<0x111, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, evt_dpp<0x112, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, evt_dpp<0x114, 0xf, 0xe>(ninf, v));
  v = fmaxf(v, evt_dpp<0x118, 0xf, 0xc>(ninf, v));
  v = fmaxf(v, evt_dpp<0x142, 0xa, 0xf>(ninf, v));
  v = fmaxf(v, evt_dpp<0x143, 0xc, 0xf>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The same reductions for NROW independent values at once, step by step across the rows, each step ONE instruction
// (v_max_f32_dpp / v_add_f32_dpp with all three operands the row's register: lanes the step does not write keep their value, which
// is what the builtin form expresses with `old` = -inf / 0).  Through __builtin_amdgcn_update_dpp + fmaxf hipcc emitted four
// instructions per max step (copy, v_mov_b32_dpp, a canonicalising v_max x, x, x, v_max) and ~90 hazard nops in the fused attention
// kernel's statistics pass.  Per row the operations and their order are those of wave_max_dpp / wave_sum_dpp: same bits.
// Hazards (inline asm is invisible to hipcc's hazard recognizer): a DPP operand needs two wait states behind the VALU write of
// its register.  Between steps the NROW >= 3 interleaved rows provide them (each row's next step is at least two VALU instructions
// behind its previous one, and nothing can be scheduled in between: the statements are volatile and ordered); in front of the FIRST
// step the producer of v[r] is compiler-scheduled code, so every row's first step carries its own `s_nop 1` INSIDE its asm string --
// the guarantee does not depend on where the scheduler puts the producer.  s_nop 1 behind the last step covers the readlane.
#define EVT_DPP_ROWS_STEP(PRE, OP, CTRL)                                                                             \
  _Pragma("unroll") for (int r = 0; r < NROW; ++r) asm volatile(PRE OP " %0, %0, %0 " CTRL : "+v"(v[r]));
#define EVT_DPP_ROWS_ALL(OP)                                                                                         \
  EVT_DPP_ROWS_STEP("s_nop 1\n\t", OP, "row_shr:1 row_mask:0xf bank_mask:0xf")                                         \
  EVT_DPP_ROWS_STEP("", OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                                                  \
  EVT_DPP_ROWS_STEP("", OP, "row_shr:4 row_mask:0xf bank_mask:0xe")                                                  \
  EVT_DPP_ROWS_STEP("", OP, "row_shr:8 row_mask:0xf bank_mask:0xc")                                                  \
  EVT_DPP_ROWS_STEP("", OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                               \
  EVT_DPP_ROWS_STEP("", OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")                                               \
  asm volatile("s_nop 1");
template <int NROW>
__device__ __forceinline__ void wave_max_dpp_rows(float* v) {
  static_assert(NROW >= 3, "interleaved rows cover the DPP wait states");
  EVT_DPP_ROWS_ALL("v_max_f32_dpp")
#pragma unroll
  for (int r = 0; r < NROW; ++r) v[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[r]), 63));
}
template <int NROW>
__device__ __forceinline__ void wave_sum_dpp_rows(float* v) {
  static_assert(NROW >= 3, "interleaved rows cover the DPP wait states");
  EVT_DPP_ROWS_ALL("v_add_f32_dpp")
#pragma unroll
  for (int r = 0; r < NROW; ++r) v[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[r]), 63));
}
#undef EVT_DPP_ROWS_ALL
#undef EVT_DPP_ROWS_STEP

// j -> (j / gw, j % gw) without an integer division: (j + 0.5) / gw is at least 0.5/gw away from every
// integer, far more than fp32 rounding for j < 2^20, gw <= 4096, so truncation is exact.
__device__ __forceinline__ int fast_div(int j, float inv_gw) { return (int)(((float)j + 0.5f) * inv_gw); }

// Storage types for tensors the reference keeps in the `matmul_2_cast` dtype (blocks.py:183-189).
// Round-to-nearest-even on store, exact widening on load.
struct bf16_t { uint16_t u; };
struct f16_t { _Float16 h; };

// fp32 -> bf16, round to nearest even, by the hardware conversion (v_cvt_pk_bf16_f32; NaN stays a quiet NaN): returns the
// rounded value widened back to fp32.  (The bit-manipulating form with its NaN branch cost ~10 instructions and two exec-mask
// branches per value: the A delta gate of the fused attention kernel spent 2.5 us per 64-column chunk on rounding.)
typedef __bf16 evt_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float evt_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_round_bits(float x, uint16_t* out) {
  union { evt_bf16x2_t b; uint32_t u; } c;
  c.b = __builtin_convertvector((evt_f32x2_t){x, x}, evt_bf16x2_t);
  *out = (uint16_t)c.u;
  return __uint_as_float(c.u << 16);
}

// round4: four values rounded at once, `out` = the stored elements (consecutive: an LDS tile quad), `back` (nullable) = the
// rounded values widened again.  bf16: two packed conversions for the four (round + store + round-again of one value at a
// time converted every element three times: 34 of the ~300 VALU instructions of a K9 gate chunk).
template <typename T> struct Store;
template <> struct Store<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  // The identity -- but OPAQUE to the compiler.  "Rounded to the store type" marks a rounding point of the reference (softmax
  // probabilities, products: blocks.py:183-189, modules.py:187-201, 285-295); with an fp32 store type nothing happens there, and hipcc
  // (-ffp-contract=fast) looked through it: `an = round(e * rinv); ad = round(an - old)` became ad = fma(e, rinv, -old), i.e. the
  // product's rounding residual instead of an exact 0 for an unchanged probability -- every frame added ~1e-9 to every element of the
  // fp32 A.v state of a clip whose input did not change (found by tests/test_gpu_properties.py: the state never reached a fixed point).
  static __device__ __forceinline__ float round(float v) { asm("" : "+v"(v)); return v; }
  static __device__ __forceinline__ void round4(const float* x, float* out, float* back) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float v = round(x[r]); out[r] = v; if (back) back[r] = v; }
  }
};
template <> struct Store<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return __uint_as_float(((uint32_t)p->u) << 16); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { uint16_t b; bf16_round_bits(v, &b); p->u = b; }
  static __device__ __forceinline__ float round(float v) { uint16_t b; return bf16_round_bits(v, &b); }
  static __device__ __forceinline__ void round4(const float* x, bf16_t* out, float* back) {
    union { evt_bf16x2_t b; uint32_t u; } c0, c1;
    c0.b = __builtin_convertvector((evt_f32x2_t){x[0], x[1]}, evt_bf16x2_t);
    c1.b = __builtin_convertvector((evt_f32x2_t){x[2], x[3]}, evt_bf16x2_t);
    out[0].u = (uint16_t)c0.u; out[1].u = (uint16_t)(c0.u >> 16);
    out[2].u = (uint16_t)c1.u; out[3].u = (uint16_t)(c1.u >> 16);
    if (back) {
      back[0] = __uint_as_float(c0.u << 16); back[1] = __uint_as_float(c0.u & 0xffff0000u);
      back[2] = __uint_as_float(c1.u << 16); back[3] = __uint_as_float(c1.u & 0xffff0000u);
    }
  }
};
template <> struct Store<f16_t> {
  static __device__ __forceinline__ float load(const f16_t* p) { return (float)p->h; }
  static __device__ __forceinline__ void store(f16_t* p, float v) { p->h = (_Float16)v; }
  static __device__ __forceinline__ float round(float v) { return (float)(_Float16)v; }
  static __device__ __forceinline__ void round4(const float* x, f16_t* out, float* back) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { out[r].h = (_Float16)x[r]; if (back) back[r] = (float)out[r].h; }
  }
};
// a * b for 0 <= a, b < 2^24 with a product below 2^32 (token index x row length): v_mul_u32_u24 is a full-rate instruction,
// the 32-bit integer multiply a quarter-rate one
__device__ __forceinline__ uint32_t evt_mul24(int a, int b) { return __umul24((uint32_t)a, (uint32_t)b); }

// Dispatch a callable templated on the storage type.
#define EVT_DISPATCH_STORE(store, T, ...)                                  \
  switch (store) {                                                          \
    case EVT_F32: { typedef float T; __VA_ARGS__; } break;                  \
    case EVT_BF16: { typedef bf16_t T; __VA_ARGS__; } break;                \
    case EVT_F16: { typedef f16_t T; __VA_ARGS__; } break;                  \
    default: return evt_fail(EVT_ERR_BAD_DTYPE, "unsupported store dtype %d", (int)(store)); \
  }

// Resolve a (group, token) pair of a window partition to a qkv-buffer row pointer (blocks.py:257-301).
// tok_map == nullptr: identity (group == clip).  Negative map entry: padding row (qkv bias).
__device__ __forceinline__ const float* evt_token_row(const float* qkv, int64_t row_stride, const int32_t* tok_map,
                                                      int groups_per_clip, int clip_rows, const float* pad_row,
                                                      int g, int t, int n_per_group) {
  if (tok_map == nullptr) return qkv + ((int64_t)g * n_per_group + t) * row_stride;
  int r = tok_map[(int64_t)(g % groups_per_clip) * n_per_group + t];
  if (r < 0) return pad_row;
  return qkv + ((int64_t)(g / groups_per_clip) * clip_rows + r) * row_stride;
}
