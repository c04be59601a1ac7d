// evt_gather.hip -- the index-structured pieces of the reference's modules that are not on the fused path, as kernels:
//
//   evt_gate_cols / evt_scatter_cols   TokenGate / TokenDeltaGate / TokenBuffer with structure="col" called stand-alone
//                                      (modules.py:90-96, 154-164, 187-201): gather / delta / reference update along the LAST axis
//   evt_gather_rows_map                rows of a (B, N, F) token tensor through an index map, with a padding row for map < 0:
//                                      `_gather_ats_skip` (blocks.py:196-203) and the window partition of the qkv buffer incl. its
//                                      padding tokens (blocks.py:257-301) in front of the window + pool path
//   evt_scatter_rows_map               the inverse for un-windowing (blocks.py:346-376): out[b][map[i]] = x[b][i], map < 0 dropped
//
// Inside EventfulBlock the same operations are fused into the attention kernels; these entry points serve the stand-alone module API
// and the rarely used window + pool combination, which ran on ATen gather / scatter / index kernels before.
#include "evt_common.h"

namespace {

// c, p: (Bp * R, N) of T; idx: (Bp, kcap); c~ / e~: (Bp * R, kcap).  One thread per (row, selected column), columns fastest.
template <typename T>
__global__ __launch_bounds__(256) void gate_cols_kernel(const T* __restrict__ c, T* p, const int32_t* __restrict__ idx,
                                                        const int32_t* __restrict__ count, int64_t total, int R, int N, int kcap,
                                                        T* __restrict__ c_tilde, T* __restrict__ e_tilde, int update_p) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t row = e / kcap;
  const int j = (int)(e - row * kcap);
  const int b = (int)(row / R);
  if (count != nullptr && j >= count[b]) return;
  const int col = idx[(int64_t)b * kcap + j];
  const T cv = c[row * N + col];
  if (c_tilde != nullptr) c_tilde[e] = cv;
  if (e_tilde != nullptr) Store<T>::store(e_tilde + e, Store<T>::load(&cv) - Store<T>::load(p + row * N + col));   // c - p in T's arithmetic
  if (update_p) p[row * N + col] = cv;
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_cols_kernel(const T* __restrict__ x, T* __restrict__ buf, const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ count, int64_t total, int R, int N, int kcap) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t row = e / kcap;
  const int j = (int)(e - row * kcap);
  const int b = (int)(row / R);
  if (count != nullptr && j >= count[b]) return;
  buf[row * N + idx[(int64_t)b * kcap + j]] = x[e];
}

// Row-structured gate / buffer for ANY element type and row length (the fp32 fast path is evt_gate_gather_update /
// evt_scatter_rows): c, p: (Bp, N, F) of T; idx (Bp, kcap); c~ / e~ (Bp, kcap, F).  One thread per element, features fastest.
template <typename T>
__global__ __launch_bounds__(256) void gate_rows_any_kernel(const T* __restrict__ c, T* p, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ count, int64_t total, int N, int F, int kcap,
                                                            T* __restrict__ c_tilde, T* __restrict__ e_tilde, int update_p) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / F;
  const int f = (int)(e - r * F);
  const int b = (int)(r / kcap), j = (int)(r - (int64_t)b * kcap);
  if (count != nullptr && j >= count[b]) return;
  const int64_t src = ((int64_t)b * N + idx[(int64_t)b * kcap + j]) * F + f;
  const T cv = c[src];
  if (c_tilde != nullptr) c_tilde[e] = cv;
  if (e_tilde != nullptr) Store<T>::store(e_tilde + e, Store<T>::load(&cv) - Store<T>::load(p + src));
  if (update_p) p[src] = cv;
}

// out[b][i][:] = x[b][map[b / rep][i]][:] (rep: consecutive batch entries sharing one index row, e.g. the heads of a clip);
// scatter = 1: out[b][map[..][i]][:] = x[b][i][:] instead.  Any element type, any row length.
template <typename T>
__global__ __launch_bounds__(256) void move_rows_any_kernel(const T* __restrict__ x, const int32_t* __restrict__ map, int64_t total, int N,
                                                            int F, int n, int rep, int scatter, T* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / F;
  const int f = (int)(e - r * F);
  const int b = (int)(r / n), i = (int)(r - (int64_t)b * n);
  const int t = map[(int64_t)(b / rep) * n + i];
  if (t < 0) return;
  if (scatter) out[((int64_t)b * N + t) * F + f] = x[e];
  else out[e] = x[((int64_t)b * N + t) * F + f];
}

// one thread per 16-byte piece of an output row
__global__ __launch_bounds__(256) void gather_rows_map_kernel(const float* __restrict__ x, const int32_t* __restrict__ map,
                                                              const float* __restrict__ pad_row, int64_t total, int N, int F4, int n_out,
                                                              int map_per_batch, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / F4;
  const int c4 = (int)(e - r * F4);
  const int b = (int)(r / n_out), i = (int)(r - (int64_t)b * n_out);
  const int src = map[(map_per_batch ? (int64_t)b * n_out : 0) + i];
  const float4 v = src >= 0 ? *reinterpret_cast<const float4*>(x + ((int64_t)b * N + src) * F4 * 4 + c4 * 4)
                            : (pad_row != nullptr ? *reinterpret_cast<const float4*>(pad_row + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
  *reinterpret_cast<float4*>(out + e * 4) = v;
}

__global__ __launch_bounds__(256) void scatter_rows_map_kernel(const float* __restrict__ x, const int32_t* __restrict__ map, int64_t total,
                                                               int n_in, int N, int F4, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / F4;
  const int c4 = (int)(e - r * F4);
  const int b = (int)(r / n_in), i = (int)(r - (int64_t)b * n_in);
  const int dst = map[i];
  if (dst < 0) return;
  *reinterpret_cast<float4*>(out + ((int64_t)b * N + dst) * F4 * 4 + c4 * 4) = *reinterpret_cast<const float4*>(x + e * 4);
}

}  // namespace

extern "C" int evt_gate_cols(const void* c, void* p, const int32_t* idx, const int32_t* count, int32_t Bp, int32_t R, int32_t N,
                             int32_t kcap, int32_t dtype, void* c_tilde, void* e_tilde, int32_t update_p, void* stream) {
  EVT_REQUIRE(c != nullptr && idx != nullptr, EVT_ERR_BAD_ARG, "evt_gate_cols: null c / idx");
  EVT_REQUIRE(p != nullptr || (!update_p && e_tilde == nullptr), EVT_ERR_BAD_ARG, "evt_gate_cols: p is null");
  EVT_REQUIRE(Bp >= 0 && R > 0 && N > 0 && kcap >= 0, EVT_ERR_BAD_ARG, "evt_gate_cols: bad sizes");
  const int64_t total = (int64_t)Bp * R * kcap;
  if (total == 0) return EVT_OK;
  EVT_REQUIRE(total < (1ll << 39), EVT_ERR_BAD_SHAPE, "%s: %lld elements exceed one launch's grid (2^31 workgroups of 256)", __func__, (long long)total);
  const dim3 grid((unsigned)((total + 255) / 256));
  EVT_DISPATCH_STORE(dtype, T, {
    hipLaunchKernelGGL(gate_cols_kernel<T>, grid, dim3(256), 0, evt_stream(stream), (const T*)c, (T*)p, idx, count, total, R, N, kcap,
                       (T*)c_tilde, (T*)e_tilde, update_p);
  });
  return evt_check_launch("evt_gate_cols");
}

extern "C" int evt_scatter_cols(const void* x, void* buf, const int32_t* idx, const int32_t* count, int32_t Bp, int32_t R, int32_t N,
                                int32_t kcap, int32_t dtype, void* stream) {
  EVT_REQUIRE(x != nullptr && buf != nullptr && idx != nullptr, EVT_ERR_BAD_ARG, "evt_scatter_cols: null pointer");
  EVT_REQUIRE(Bp >= 0 && R > 0 && N > 0 && kcap >= 0, EVT_ERR_BAD_ARG, "evt_scatter_cols: bad sizes");
  const int64_t total = (int64_t)Bp * R * kcap;
  if (total == 0) return EVT_OK;
  EVT_REQUIRE(total < (1ll << 39), EVT_ERR_BAD_SHAPE, "%s: %lld elements exceed one launch's grid (2^31 workgroups of 256)", __func__, (long long)total);
  const dim3 grid((unsigned)((total + 255) / 256));
  EVT_DISPATCH_STORE(dtype, T, {
    hipLaunchKernelGGL(scatter_cols_kernel<T>, grid, dim3(256), 0, evt_stream(stream), (const T*)x, (T*)buf, idx, count, total, R, N, kcap);
  });
  return evt_check_launch("evt_scatter_cols");
}

extern "C" int evt_gather_rows_map(const float* x, const int32_t* map, const float* pad_row, int32_t B, int32_t N, int32_t F, int32_t n_out,
                                   int32_t map_per_batch, float* out, void* stream) {
  EVT_REQUIRE(x != nullptr && map != nullptr && out != nullptr, EVT_ERR_BAD_ARG, "evt_gather_rows_map: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0 && F > 0 && n_out >= 0, EVT_ERR_BAD_ARG, "evt_gather_rows_map: bad sizes");
  EVT_REQUIRE((F & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_gather_rows_map: F=%d must be a multiple of 4", F);
  const int64_t total = (int64_t)B * n_out * (F / 4);
  if (total == 0) return EVT_OK;
  EVT_REQUIRE(total < (1ll << 39), EVT_ERR_BAD_SHAPE, "%s: %lld elements exceed one launch's grid (2^31 workgroups of 256)", __func__, (long long)total);
  hipLaunchKernelGGL(gather_rows_map_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, evt_stream(stream), x, map, pad_row, total, N,
                     F / 4, n_out, map_per_batch, out);
  return evt_check_launch("evt_gather_rows_map");
}

extern "C" int evt_scatter_rows_map(const float* x, const int32_t* map, int32_t B, int32_t n_in, int32_t N, int32_t F, float* out, void* stream) {
  EVT_REQUIRE(x != nullptr && map != nullptr && out != nullptr, EVT_ERR_BAD_ARG, "evt_scatter_rows_map: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0 && F > 0 && n_in >= 0, EVT_ERR_BAD_ARG, "evt_scatter_rows_map: bad sizes");
  EVT_REQUIRE((F & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_scatter_rows_map: F=%d must be a multiple of 4", F);
  const int64_t total = (int64_t)B * n_in * (F / 4);
  if (total == 0) return EVT_OK;
  EVT_REQUIRE(total < (1ll << 39), EVT_ERR_BAD_SHAPE, "%s: %lld elements exceed one launch's grid (2^31 workgroups of 256)", __func__, (long long)total);
  hipLaunchKernelGGL(scatter_rows_map_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, evt_stream(stream), x, map, total, n_in, N,
                     F / 4, out);
  return evt_check_launch("evt_scatter_rows_map");
}

extern "C" int evt_gate_rows_any(const void* c, void* p, const int32_t* idx, const int32_t* count, int32_t Bp, int32_t N, int32_t F,
                                 int32_t kcap, int32_t dtype, void* c_tilde, void* e_tilde, int32_t update_p, void* stream) {
  EVT_REQUIRE(c != nullptr && idx != nullptr, EVT_ERR_BAD_ARG, "evt_gate_rows_any: null c / idx");
  EVT_REQUIRE(p != nullptr || (!update_p && e_tilde == nullptr), EVT_ERR_BAD_ARG, "evt_gate_rows_any: p is null");
  EVT_REQUIRE(Bp >= 0 && N > 0 && F > 0 && kcap >= 0, EVT_ERR_BAD_ARG, "evt_gate_rows_any: bad sizes");
  const int64_t total = (int64_t)Bp * kcap * F;
  if (total == 0) return EVT_OK;
  EVT_REQUIRE(total < (1ll << 39), EVT_ERR_BAD_SHAPE, "%s: %lld elements exceed one launch's grid (2^31 workgroups of 256)", __func__, (long long)total);
  const dim3 grid((unsigned)((total + 255) / 256));
  EVT_DISPATCH_STORE(dtype, T, {
    hipLaunchKernelGGL(gate_rows_any_kernel<T>, grid, dim3(256), 0, evt_stream(stream), (const T*)c, (T*)p, idx, count, total, N, F, kcap,
                       (T*)c_tilde, (T*)e_tilde, update_p);
  });
  return evt_check_launch("evt_gate_rows_any");
}

extern "C" int evt_move_rows_any(const void* x, const int32_t* map, int32_t B, int32_t N, int32_t F, int32_t n, int32_t rep, int32_t scatter,
                                 int32_t dtype, void* out, void* stream) {
  EVT_REQUIRE(x != nullptr && map != nullptr && out != nullptr, EVT_ERR_BAD_ARG, "evt_move_rows_any: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0 && F > 0 && n >= 0 && rep > 0, EVT_ERR_BAD_ARG, "evt_move_rows_any: bad sizes");
  const int64_t total = (int64_t)B * n * F;
  if (total == 0) return EVT_OK;
  EVT_REQUIRE(total < (1ll << 39), EVT_ERR_BAD_SHAPE, "%s: %lld elements exceed one launch's grid (2^31 workgroups of 256)", __func__, (long long)total);
  const dim3 grid((unsigned)((total + 255) / 256));
  EVT_DISPATCH_STORE(dtype, T, {
    hipLaunchKernelGGL(move_rows_any_kernel<T>, grid, dim3(256), 0, evt_stream(stream), (const T*)x, map, total, N, F, n, rep, scatter, (T*)out);
  });
  return evt_check_launch("evt_move_rows_any");
}
