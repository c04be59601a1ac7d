// evt_rows.hip -- HBM-bound per-token passes: residual + LayerNorm + gate delta-norm (K0/K1a),
// gate gather / reference update (K2), token-buffer row scatter.
//
// Layout: one 64-lane wavefront owns one token row; lanes read float4 at stride 64 so every
// wave-instruction moves 1 KiB contiguous.  No LDS: the row lives in registers between the
// statistics pass and the normalise / norm pass, so each input byte is read from HBM exactly once.
#include "evt_common.h"

namespace {

template <int NV>
__global__ __launch_bounds__(256) void row_pass_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       int res_rows, float* __restrict__ sum_out, const float* __restrict__ ln_w,
                                                       const float* __restrict__ ln_b, float eps,
                                                       float* __restrict__ c_out, const float* __restrict__ p,
                                                       float* __restrict__ norms, int rows, int D, int order) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const int64_t rbase = (int64_t)(res_rows > 0 ? row % res_rows : row) * D;  // broadcast over clips
  const int nvec = D >> 2;
  // Every operand of the row is requested up front, through clamped addresses (no per-element branch, so the loads
  // of all NV chunks are in flight together): x, the residual, the gate reference p -- which does not depend on the
  // LayerNorm result -- and the LN weights.  The statistics then run while the later loads are still landing.
  const bool want_norm = norms != nullptr;
  const bool has_p = want_norm && p != nullptr;
  int cc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) cc[i] = min(lane + i * 64, nvec - 1) * 4;
  float4 v[NV], r[NV], pr[NV], lw[NV], lb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(x + base + cc[i]);
  if (has_p) {
#pragma unroll
    for (int i = 0; i < NV; ++i) pr[i] = *reinterpret_cast<const float4*>(p + base + cc[i]);
  }
  if (ln_w != nullptr) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      lw[i] = *reinterpret_cast<const float4*>(ln_w + cc[i]);
      lb[i] = *reinterpret_cast<const float4*>(ln_b + cc[i]);
    }
  }
  if (res != nullptr) {   // last: hipcc merges the residual adds (and their wait) into this block
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = *reinterpret_cast<const float4*>(res + rbase + cc[i]);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bool in = lane + i * 64 < nvec;
    if (res != nullptr) { v[i].x += r[i].x; v[i].y += r[i].y; v[i].z += r[i].z; v[i].w += r[i].w; }
    if (!in) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sum_out != nullptr && in) *reinterpret_cast<float4*>(sum_out + base + cc[i]) = v[i];
  }
  if (ln_w != nullptr) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + i * 64 < nvec) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x = (v[i].x - mean) * rstd * lw[i].x + lb[i].x;
      v[i].y = (v[i].y - mean) * rstd * lw[i].y + lb[i].y;
      v[i].z = (v[i].z - mean) * rstd * lw[i].z + lb[i].z;
      v[i].w = (v[i].w - mean) * rstd * lw[i].w + lb[i].w;
    }
  }
  if (c_out != nullptr) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nvec) *reinterpret_cast<float4*>(c_out + base + cc[i]) = v[i];
  }
  if (want_norm) {   // order (policies.py:11,44,76: `vector_norm(x, ord=order)`): 2 -> sqrt(sum of squares), 1 -> sum |.|, 0 -> max |.| (inf)
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + i * 64 < nvec) {
        const float4 z = has_p ? pr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float a = v[i].x - z.x, b = v[i].y - z.y, c = v[i].z - z.z, d = v[i].w - z.w;
        if (order == 2) q += (a * a + b * b) + (c * c + d * d);
        else if (order == 1) q += (fabsf(a) + fabsf(b)) + (fabsf(c) + fabsf(d));
        else q = fmaxf(q, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
      }
    }
    q = order == 0 ? wave_max(q) : wave_sum(q);
    if (lane == 0) norms[row] = order == 2 ? sqrtf(q) : q;
  }
}

// One wave per selected row (b, i).
__global__ __launch_bounds__(256) void gather_update_kernel(const float* __restrict__ c, float* __restrict__ p,
                                                            const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ count, int B, int N, int D,
                                                            int kcap, float* __restrict__ c_tilde,
                                                            float* __restrict__ e_tilde, int update_p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= B * kcap) return;
  const int b = m / kcap, i = m - b * kcap;
  if (count != nullptr && i >= count[b]) return;
  const int tok = idx[m];
  const int64_t src = ((int64_t)b * N + tok) * D;
  const int64_t dst = (int64_t)m * D;
  for (int c4 = lane; c4 < (D >> 2); c4 += 64) {
    const float4 v = *reinterpret_cast<const float4*>(c + src + c4 * 4);
    if (c_tilde != nullptr) *reinterpret_cast<float4*>(c_tilde + dst + c4 * 4) = v;
    if (e_tilde != nullptr) {
      const float4 r = *reinterpret_cast<const float4*>(p + src + c4 * 4);
      *reinterpret_cast<float4*>(e_tilde + dst + c4 * 4) = make_float4(v.x - r.x, v.y - r.y, v.z - r.z, v.w - r.w);
    }
    if (update_p) *reinterpret_cast<float4*>(p + src + c4 * 4) = v;
  }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ x, float* __restrict__ buf,
                                                           const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ count, int B, int N, int F,
                                                           int kcap) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= B * kcap) return;
  const int b = m / kcap, i = m - b * kcap;
  if (count != nullptr && i >= count[b]) return;
  const int64_t dst = ((int64_t)b * N + idx[m]) * F;
  const int64_t src = (int64_t)m * F;
  for (int c4 = lane; c4 < (F >> 2); c4 += 64)
    *reinterpret_cast<float4*>(buf + dst + c4 * 4) = *reinterpret_cast<const float4*>(x + src + c4 * 4);
}

}  // namespace

extern "C" int evt_row_pass_ord(const float* x, const float* res, int res_rows, float* sum_out, const float* ln_w,
                                const float* ln_b, float eps, float* c_out, const float* p, float* norms, int rows, int D,
                                int order, void* stream) {
  EVT_REQUIRE(order == EVT_NORM_L2 || order == EVT_NORM_L1 || order == EVT_NORM_LINF, EVT_ERR_BAD_ARG, "evt_row_pass: norm order %d", order);
  EVT_REQUIRE(x != nullptr, EVT_ERR_BAD_ARG, "evt_row_pass: x is null");
  EVT_REQUIRE(rows >= 0 && D > 0, EVT_ERR_BAD_ARG, "evt_row_pass: rows=%d D=%d", rows, D);
  EVT_REQUIRE((D & 3) == 0 && D <= 4096, EVT_ERR_BAD_SHAPE, "evt_row_pass: D=%d must be a multiple of 4 and <= 4096", D);
  EVT_REQUIRE((ln_w == nullptr) == (ln_b == nullptr), EVT_ERR_BAD_ARG, "evt_row_pass: ln_w and ln_b must come together");
  EVT_REQUIRE(p == nullptr || norms != nullptr, EVT_ERR_BAD_ARG, "evt_row_pass: p given without norms");
  EVT_REQUIRE(res_rows >= 0, EVT_ERR_BAD_ARG, "evt_row_pass: res_rows=%d", res_rows);
  if (rows == 0) return EVT_OK;
  const dim3 grid((rows + 3) / 4), block(256);
  const int need = (D / 4 + 63) / 64;
  hipStream_t s = evt_stream(stream);
#define LAUNCH(NV) hipLaunchKernelGGL(row_pass_kernel<NV>, grid, block, 0, s, x, res, res_rows, sum_out, ln_w, ln_b, eps, c_out, p, norms, rows, D, order)
  if (need <= 1) LAUNCH(1);
  else if (need <= 2) LAUNCH(2);
  else if (need <= 3) LAUNCH(3);
  else if (need <= 4) LAUNCH(4);
  else if (need <= 8) LAUNCH(8);
  else LAUNCH(16);
#undef LAUNCH
  return evt_check_launch("evt_row_pass");
}

extern "C" int evt_row_pass(const float* x, const float* res, int res_rows, float* sum_out, const float* ln_w,
                            const float* ln_b, float eps, float* c_out, const float* p, float* norms, int rows, int D,
                            void* stream) {
  return evt_row_pass_ord(x, res, res_rows, sum_out, ln_w, ln_b, eps, c_out, p, norms, rows, D, EVT_NORM_L2, stream);
}

extern "C" int evt_gate_gather_update(const float* c, float* p, const int32_t* idx, const int32_t* count, int B, int N,
                                      int D, int kcap, float* c_tilde, float* e_tilde, int update_p, void* stream) {
  EVT_REQUIRE(c != nullptr && idx != nullptr, EVT_ERR_BAD_ARG, "evt_gate_gather_update: null c/idx");
  EVT_REQUIRE(p != nullptr || (!update_p && e_tilde == nullptr), EVT_ERR_BAD_ARG, "evt_gate_gather_update: p is null");
  EVT_REQUIRE(B >= 0 && N > 0 && D > 0 && kcap >= 0, EVT_ERR_BAD_ARG, "evt_gate_gather_update: bad sizes");
  EVT_REQUIRE((D & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_gate_gather_update: D=%d must be a multiple of 4", D);
  if (B * kcap == 0) return EVT_OK;
  hipLaunchKernelGGL(gather_update_kernel, dim3((B * kcap + 3) / 4), dim3(256), 0, evt_stream(stream), c, p, idx,
                     count, B, N, D, kcap, c_tilde, e_tilde, update_p);
  return evt_check_launch("evt_gate_gather_update");
}

extern "C" int evt_scatter_rows(const float* x, float* buf, const int32_t* idx, const int32_t* count, int B, int N,
                                int F, int kcap, void* stream) {
  EVT_REQUIRE(x != nullptr && buf != nullptr && idx != nullptr, EVT_ERR_BAD_ARG, "evt_scatter_rows: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0 && F > 0 && kcap >= 0, EVT_ERR_BAD_ARG, "evt_scatter_rows: bad sizes");
  EVT_REQUIRE((F & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_scatter_rows: F=%d must be a multiple of 4", F);
  if (B * kcap == 0) return EVT_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((B * kcap + 3) / 4), dim3(256), 0, evt_stream(stream), x, buf, idx,
                     count, B, N, F, kcap);
  return evt_check_launch("evt_scatter_rows");
}
