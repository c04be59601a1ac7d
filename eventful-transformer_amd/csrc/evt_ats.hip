// evt_ats.hip -- adaptive token sampling (Block._adaptive_token_sampling / _stabilize_ats_indices, blocks.py:150-181, 378-391):
//
//   evt_ats_scores     raw[b,h,n] = a[b,h,n,0] * || v[b,h,n,:] ||;  s[b,h,n] = raw[b,h,n] / sum_{n' >= 1} raw[b,h,n'];  s[b,h,0] = inf;
//                      scores[h,n] = sum_b s[b,h,n]   -- the reference reduces dim -3 of a (B,H,N) tensor, i.e. the BATCH axis
//                      (blocks.py:163), which only type-checks downstream when batch == heads: clip b then keeps the tokens ranked by
//                      row b of `scores`.  Every intermediate is rounded to the tensors' dtype where the reference's op sequence
//                      rounds it (the EventfulBlock path scores AFTER the matmul_2 cast, blocks.py:561-562): norm, product, row sum,
//                      quotient, batch sum.
//   (selection)        evt_select_topk on the (H, N) scores: ascending index lists (the reference sorts them, blocks.py:380).
//   evt_ats_stabilize  keep every surviving token at last frame's position: out = last, with the entries of `last` that are not in
//                      `now` replaced, in order, by the entries of `now` that are not in `last` (both ascending; blocks.py:378-391).
//
// Off the gated fast path (a baseline the paper compares against); the kernels replace ~15 ATen launches per block and frame.
#include "evt_common.h"

namespace {

constexpr int ATS_MAX_PER_THREAD = 16;   // N <= 4096

template <typename T>
__global__ __launch_bounds__(256) void ats_scores_kernel(const T* __restrict__ a, const T* __restrict__ v, int64_t v_bs, int64_t v_hs,
                                                         int64_t v_rs, int B, int H, int N, int dh, float* __restrict__ out) {
  __shared__ float red[256];
  const int h = blockIdx.x, tid = threadIdx.x;
  float acc[ATS_MAX_PER_THREAD], raw[ATS_MAX_PER_THREAD];
#pragma unroll
  for (int i = 0; i < ATS_MAX_PER_THREAD; ++i) acc[i] = 0.f;
  for (int b = 0; b < B; ++b) {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < ATS_MAX_PER_THREAD; ++i) {
      const int n = tid + 256 * i;
      raw[i] = 0.f;
      if (n < N) {
        const T* vr = v + b * v_bs + h * v_hs + n * v_rs;
        float ss = 0.f;
        for (int d = 0; d < dh; ++d) { const float x = Store<T>::load(vr + d); ss = fmaf(x, x, ss); }
        const float nv = Store<T>::round(sqrtf(ss));
        raw[i] = Store<T>::round(Store<T>::load(a + (((int64_t)b * H + h) * N + n) * N) * nv);
        if (n >= 1) part += raw[i];
      }
    }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    const float S = Store<T>::round(red[0]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ATS_MAX_PER_THREAD; ++i) acc[i] += Store<T>::round(raw[i] / S);
  }
#pragma unroll
  for (int i = 0; i < ATS_MAX_PER_THREAD; ++i) {
    const int n = tid + 256 * i;
    if (n < N) out[(int64_t)h * N + n] = n == 0 ? INFINITY : Store<T>::round(acc[i]);   // the class token always stays (blocks.py:161)
  }
}

__global__ __launch_bounds__(256) void ats_stabilize_kernel(const int32_t* __restrict__ last, const int32_t* __restrict__ now, int n, int N,
                                                            int32_t* __restrict__ out) {
  extern __shared__ uint32_t ats_bits[];   // [2][ceil(N / 32)]: membership of `now`, of `last`
  const int words = (N + 31) >> 5, row = blockIdx.x, tid = threadIdx.x;
  uint32_t* in_now = ats_bits;
  uint32_t* in_last = ats_bits + words;
  const int32_t* L = last + (int64_t)row * n;
  const int32_t* W = now + (int64_t)row * n;
  for (int i = tid; i < 2 * words; i += 256) ats_bits[i] = 0u;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    atomicOr(in_now + (W[i] >> 5), 1u << (W[i] & 31));
    atomicOr(in_last + (L[i] >> 5), 1u << (L[i] & 31));
  }
  __syncthreads();
  if (tid == 0) {   // (a few thousand steps at most, once per block and frame of a baseline path)
    int j = 0;
    for (int i = 0; i < n; ++i) {
      const int t = L[i];
      if ((in_now[t >> 5] >> (t & 31)) & 1u) { out[(int64_t)row * n + i] = t; continue; }
      while (j < n && ((in_last[W[j] >> 5] >> (W[j] & 31)) & 1u)) ++j;
      out[(int64_t)row * n + i] = j < n ? W[j++] : t;
    }
  }
}

}  // namespace

extern "C" int evt_ats_scores(const void* a, const void* v, int64_t v_bs, int64_t v_hs, int64_t v_rs, int32_t B, int32_t H, int32_t N,
                              int32_t dh, int32_t dtype, float* scores, void* stream) {
  EVT_REQUIRE(a != nullptr && v != nullptr && scores != nullptr, EVT_ERR_BAD_ARG, "evt_ats_scores: null pointer");
  EVT_REQUIRE(B > 0 && H > 0 && N > 1 && dh > 0, EVT_ERR_BAD_ARG, "evt_ats_scores: bad sizes");
  EVT_REQUIRE(N <= 256 * ATS_MAX_PER_THREAD, EVT_ERR_BAD_SHAPE, "evt_ats_scores: N=%d exceeds %d tokens", N, 256 * ATS_MAX_PER_THREAD);
  EVT_DISPATCH_STORE(dtype, T, {
    hipLaunchKernelGGL(ats_scores_kernel<T>, dim3(H), dim3(256), 0, evt_stream(stream), (const T*)a, (const T*)v, v_bs, v_hs, v_rs, B, H, N, dh,
                       scores);
  });
  return evt_check_launch("evt_ats_scores");
}

extern "C" int evt_ats_stabilize(const int32_t* last, const int32_t* now, int32_t rows, int32_t n, int32_t N, int32_t* out, void* stream) {
  EVT_REQUIRE(last != nullptr && now != nullptr && out != nullptr, EVT_ERR_BAD_ARG, "evt_ats_stabilize: null pointer");
  EVT_REQUIRE(rows >= 0 && n >= 0 && N > 0 && n <= N, EVT_ERR_BAD_ARG, "evt_ats_stabilize: bad sizes");
  if (rows == 0 || n == 0) return EVT_OK;
  const size_t lds = (size_t)2 * ((N + 31) / 32) * 4;
  hipLaunchKernelGGL(ats_stabilize_kernel, dim3(rows), dim3(256), lds, evt_stream(stream), last, now, n, N, out);
  return evt_check_launch("evt_ats_stabilize");
}
