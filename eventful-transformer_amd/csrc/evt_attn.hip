// evt_attn.hip -- K4 (q.k^T product state), K5 (softmax + rel-pos + attention delta gate),
// K6a (value delta gate), K6 (attention-value product state).
//
// All contractions here run on fp32-input MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32).  Values the
// reference holds in bf16/fp16 after `_cast_matmul_2` are widened exactly to fp32 when staged, so
// products are exact and accumulation is fp32 -- the same arithmetic a bf16 MFMA would do --
// and every point where the reference rounds to the cast dtype is reproduced by Store<T>::round.
#include "evt_common.h"
#include "evt_prep_roles.h"

namespace {

// =============================================================================================
// K4: product[g,h,rm(m),cn(n)] = sum_d (q[rm(m)][d] / scale) * k[cn(n)][d]
//   part 0 (FULL)      : rm, cn identity                         (modules.py:224-230)
//   part 1 (DELTA rows): rm = idx_q, cn identity                  (modules.py:236-241)
//   part 2 (DELTA cols): rm identity, cn = idx_k                  (modules.py:242-247)
// 128x128 output tile per workgroup (2x2 waves of 2x2 32x32 MFMA accumulators); the head dim is staged
// through LDS in chunks of 32 with register prefetch of the next chunk under the MFMAs.
// q / k are addressed as base + clip*bs + token*rs + head*hs so both the packed (B,N,3D) token
// buffer of the blocks and free-standing (B,H,N,dh) tensors fit.
// =============================================================================================
struct QkArgs {
  const float* q; int64_t q_bs, q_hs, q_rs;
  const float* k; int64_t k_bs, k_hs, k_rs;
  float* product;
  const int32_t* idx_q; const int32_t* count_q; int kcap_q; const int32_t* idx_q_rest;
  const int32_t* idx_k; const int32_t* count_k; int kcap_k;
  const int32_t* tok_map; int groups_per_clip; const float* pad_q; const float* pad_k;
  int G, H, Nq, Nk, dh;
  float scale;
  int delta;  // 0: full; 1: rows+cols delta (blockIdx.z selects the part)
  int split;  // 1: bf16 hi/lo split-precision MFMA
};

__device__ __forceinline__ const float* qk_row(const float* base, int64_t bs, int64_t rs, const int32_t* tok_map,
                                               int gpc, const float* pad, int g, int t, int n_per_group) {
  if (tok_map == nullptr) return base + (int64_t)g * bs + (int64_t)t * rs;
  const int r = tok_map[(int64_t)(g % gpc) * n_per_group + t];
  if (r < 0) return pad;
  return base + (int64_t)(g / gpc) * bs + (int64_t)r * rs;
}

constexpr int QT = 128;   // output tile edge
constexpr int QKC = 32;   // head-dim chunk staged per pass (LDS 2 x 128 x 36 floats = 36 KB -> 4 workgroups / CU)
constexpr int QLD = QKC + 4;

typedef __bf16 qk_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 qk_bf16x4 __attribute__((ext_vector_type(4)));
typedef float qk_f32x4 __attribute__((ext_vector_type(4)));
constexpr int QSP = QKC + 8;  // bf16 pitch of the split planes (80 bytes: conflict-free 16-byte reads)

// x = hi + lo, hi = rne_bf16(x), lo = rne_bf16(x - hi): the split-precision operands of evt_linear.hip
__device__ __forceinline__ void qk_split4(const float4 v, qk_bf16x4* hi, qk_bf16x4* lo) {
  const qk_f32x4 x = {v.x, v.y, v.z, v.w};
  const qk_bf16x4 hh = __builtin_convertvector(x, qk_bf16x4);
  *hi = hh;
  *lo = __builtin_convertvector(x - __builtin_convertvector(hh, qk_f32x4), qk_bf16x4);
}

// SPLIT = false: fp32-input MFMA (exact fp32 products).  SPLIT = true: q and k are split into bf16 hi/lo planes while
// they are staged and (q/scale).k = lo.hi + hi.lo + hi.hi runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (~1e-5 relative, K = head dim only; 5x less matrix-pipe time than the 32x32x2 fp32 MFMA that bounds this kernel).
template <bool SPLIT>
__global__ __launch_bounds__(256, 3) void qk_kernel(const QkArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_raw[SPLIT ? (4 * QT * QSP) / 2 : 2 * QT * QLD];
  float* As = lds_raw;
  float* Bs = lds_raw + QT * QLD;
  __bf16* Ahi = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Alo = Ahi + QT * QSP;
  __bf16* Bhi = Alo + QT * QSP;
  __bf16* Blo = Bhi + QT * QSP;
  __shared__ int rmap[QT];
  __shared__ int cmap[QT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int z = blockIdx.z;
  int part = 0;
  if (a.delta) { part = 1 + (z & 1); z >>= 1; }
  const int g = z / a.H, h = z - g * a.H;
  const int b = a.tok_map ? g / a.groups_per_clip : g;
  int m_lim, n_lim, m0, n0;
  if (part == 2) {
    m0 = blockIdx.x * QT; n0 = blockIdx.y * QT;
    // rows already rewritten whole by part 1 are skipped when the complement list is available
    m_lim = a.idx_q_rest ? a.Nq - (a.count_q ? a.count_q[b] : a.kcap_q) : a.Nq;
    n_lim = a.count_k ? a.count_k[b] : a.kcap_k;
  } else {
    m0 = blockIdx.y * QT; n0 = blockIdx.x * QT;
    m_lim = (part == 1) ? (a.count_q ? a.count_q[b] : a.kcap_q) : a.Nq; n_lim = a.Nk;
  }
  if (m0 >= m_lim || n0 >= n_lim) return;

  // staging assignment: thread -> rows r0 + 32 j (j = 0..3), one float4 at column c4 * 4 of the chunk.
  // Index lists and the window map are read with clamped addresses inside UNIFORM branches, so the 4 loads of a
  // list are in flight together (a per-element `cond ? list[m] : -1` compiles to load + s_waitcnt vmcnt(0) each:
  // ~20 serialised round trips before the first MFMA).
  const int r0 = tid >> 3, c4 = tid & 7;
  const int32_t* msrc = (part == 1) ? a.idx_q + (int64_t)b * a.kcap_q
                      : (part == 2 && a.idx_q_rest) ? a.idx_q_rest + (int64_t)b * a.Nq : nullptr;
  const int32_t* nsrc = (part == 2) ? a.idx_k + (int64_t)b * a.kcap_k : nullptr;
  int tmv[4], tnv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + r0 + 32 * j, n = n0 + r0 + 32 * j;
    tmv[j] = m < m_lim ? m : m_lim - 1;   // m0 < m_lim, n0 < n_lim: the clamps are valid positions
    tnv[j] = n < n_lim ? n : n_lim - 1;
  }
  if (msrc != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tmv[j] = msrc[tmv[j]];
  }
  if (nsrc != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tnv[j] = nsrc[tnv[j]];
  }
  int rqv[4], rkv[4];   // clip rows (window map applied), -1 = padding vector
#pragma unroll
  for (int j = 0; j < 4; ++j) { rqv[j] = tmv[j]; rkv[j] = tnv[j]; }
  if (a.tok_map != nullptr) {
    const int32_t* wmap_q = a.tok_map + (int64_t)(g % a.groups_per_clip) * a.Nq;
    const int32_t* wmap_k = a.tok_map + (int64_t)(g % a.groups_per_clip) * a.Nk;
#pragma unroll
    for (int j = 0; j < 4; ++j) { rqv[j] = wmap_q[tmv[j]]; rkv[j] = wmap_k[tnv[j]]; }
  }
  const int64_t clip = a.tok_map ? (int64_t)(g / a.groups_per_clip) : (int64_t)g;
  const float* qp[4];
  const float* kp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + 32 * j;
    const bool mv = (m0 + r) < m_lim, nv = (n0 + r) < n_lim;
    if (c4 == 0) { rmap[r] = mv ? tmv[j] : -1; cmap[r] = nv ? tnv[j] : -1; }
    qp[j] = !mv ? nullptr : (rqv[j] < 0 ? a.pad_q : a.q + clip * a.q_bs + (int64_t)rqv[j] * a.q_rs) + h * a.q_hs;
    kp[j] = !nv ? nullptr : (rkv[j] < 0 ? a.pad_k : a.k + clip * a.k_bs + (int64_t)rkv[j] * a.k_rs) + h * a.k_hs;
  }
  // q / self.scale (blocks.py:514): a power-of-two scale (dh = 16, 64, 256) makes x * (1/scale) exact
  const float inv = 1.0f / a.scale;
  const bool pow2 = (inv * a.scale == 1.0f) && ((__float_as_uint(a.scale) & 0x007fffffu) == 0u);

  float4 rq[4], rk[4];
  auto fetch = [&](int d0) {
    const int d = d0 + c4 * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rq[j] = (qp[j] != nullptr && d < a.dh) ? *reinterpret_cast<const float4*>(qp[j] + d) : make_float4(0.f, 0.f, 0.f, 0.f);
      rk[j] = (kp[j] != nullptr && d < a.dh) ? *reinterpret_cast<const float4*>(kp[j] + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  // 2 x 2 waves, each a 64 x 64 region = 2 x 2 accumulators of 32 x 32; sub-tiles wholly outside the
  // valid range are skipped (N = 197 -> 7 of 8 32-wide column groups are live).
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  bool live_m[2], live_n[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    live_m[i] = (m0 + wm * 64 + i * 32) < m_lim;
    live_n[i] = (n0 + wn * 64 + i * 32) < n_lim;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  for (int d0 = 0; d0 < a.dh; d0 += QKC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 qa = rq[j];
      if (pow2) { qa.x *= inv; qa.y *= inv; qa.z *= inv; qa.w *= inv; }
      else { qa.x /= a.scale; qa.y /= a.scale; qa.z /= a.scale; qa.w /= a.scale; }
      if (SPLIT) {
        qk_bf16x4 hq, lq, hk, lk;
        qk_split4(qa, &hq, &lq);
        qk_split4(rk[j], &hk, &lk);
        const int o = (r0 + 32 * j) * QSP + c4 * 4;
        *reinterpret_cast<qk_bf16x4*>(Ahi + o) = hq;
        *reinterpret_cast<qk_bf16x4*>(Alo + o) = lq;
        *reinterpret_cast<qk_bf16x4*>(Bhi + o) = hk;
        *reinterpret_cast<qk_bf16x4*>(Blo + o) = lk;
      } else {
        *reinterpret_cast<float4*>(As + (r0 + 32 * j) * QLD + c4 * 4) = qa;
        *reinterpret_cast<float4*>(Bs + (r0 + 32 * j) * QLD + c4 * 4) = rk[j];
      }
    }
    __syncthreads();
    if (d0 + QKC < a.dh) fetch(d0 + QKC);
    if (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < QKC; ks += 16) {
        qk_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ao = (wm * 64 + i * 32 + lr) * QSP + ks + 8 * lh;
          const int bo = (wn * 64 + i * 32 + lr) * QSP + ks + 8 * lh;
          ah[i] = *reinterpret_cast<const qk_bf16x8*>(Ahi + ao);
          al[i] = *reinterpret_cast<const qk_bf16x8*>(Alo + ao);
          bh[i] = *reinterpret_cast<const qk_bf16x8*>(Bhi + bo);
          bl[i] = *reinterpret_cast<const qk_bf16x8*>(Blo + bo);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (!(live_m[i] && live_n[j])) continue;  // wave-uniform
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    } else {
    // lane half lh covers k in [16 lh, 16 lh + 16) of the chunk (k order is free inside a sum)
#pragma unroll
    for (int q = 0; q < QKC / 2; q += 4) {
      float4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const float4*>(As + (wm * 64 + i * 32 + lr) * QLD + lh * (QKC / 2) + q);
        fb[i] = *reinterpret_cast<const float4*>(Bs + (wn * 64 + i * 32 + lr) * QLD + lh * (QKC / 2) + q);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (!(live_m[i] && live_n[j])) continue;  // wave-uniform
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    }
    __syncthreads();
  }
  float* out = a.product + ((int64_t)g * a.H + h) * a.Nq * a.Nk;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (!live_n[j]) continue;
    const int cn = cmap[wn * 64 + j * 32 + lr];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!live_m[i]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rm = rmap[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        if (rm >= 0 && cn >= 0) out[(int64_t)rm * a.Nk + cn] = acc[i][j][r];
      }
    }
  }
}

int launch_qk(const QkArgs& a, void* stream) {
  const int ntq = (a.Nq + QT - 1) / QT, ntk = (a.Nk + QT - 1) / QT;
  dim3 grid;
  if (a.delta) {
    // part 1: x -> Nk tiles, y -> kcap_q tiles; part 2: x -> Nq tiles, y -> kcap_k tiles
    const int gx = ntq > ntk ? ntq : ntk;
    const int ky = ((a.kcap_q > a.kcap_k ? a.kcap_q : a.kcap_k) + QT - 1) / QT;
    grid = dim3(gx, ky, a.G * a.H * 2);
  } else {
    grid = dim3(ntk, ntq, a.G * a.H);
  }
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) return EVT_OK;
  if (a.split) hipLaunchKernelGGL(qk_kernel<true>, grid, dim3(256), 0, evt_stream(stream), a);
  else hipLaunchKernelGGL(qk_kernel<false>, grid, dim3(256), 0, evt_stream(stream), a);
  return evt_check_launch("evt_qk");
}

// =============================================================================================
// K5: one wavefront per attention row.
// =============================================================================================
struct SmArgs {
  const float* product; const float* qkv; const float* rel_y; const float* rel_x;
  const int32_t* tok_map; const float* pad_row; int groups_per_clip, clip_rows;
  void* a_state; void* a_new; void* a_delta;
  const int32_t* idx; const int32_t* count;
  int G, H, N, Nk, D, dh, kcap, gh, gw, qw, gated;   // gh x gw: KEY grid; qw: query grid width
};

template <typename T>
__global__ __launch_bounds__(256) void softmax_gate_kernel(const SmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;      // (g*H + h)*N + i
  const int64_t total = (int64_t)a.G * a.H * a.N;
  if (row >= total) return;
  const int i = (int)(row % a.N);
  const int64_t gh_ = row / a.N;
  const int h = (int)(gh_ % a.H), g = (int)(gh_ / a.H);
  const int b = a.tok_map ? g / a.groups_per_clip : g;
  const int per_wave = a.Nk + a.dh + a.gh + a.gw;
  float* lg = smem + (size_t)wave * per_wave;  // logits, then exp
  float* qs = lg + a.Nk;
  float* ry = qs + a.dh;
  float* rx = ry + a.gh;

  const bool rel = a.rel_y != nullptr;
  const float inv_gw = rel ? 1.0f / (float)a.gw : 0.f;
  if (rel) {
    const float* qrow = evt_token_row(a.qkv, 3 * (int64_t)a.D, a.tok_map, a.groups_per_clip, a.clip_rows, a.pad_row,
                                      g, i, a.N) + h * a.dh;
    for (int d = lane; d < a.dh; d += 64) qs[d] = qrow[d];
    // same-wave LDS visibility: wave-synchronous, but make the compiler keep the order
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const int yi = i / a.qw, xi = i - yi * a.qw;
    for (int e = lane; e < a.gh + a.gw; e += 64) {
      const float* tab = (e < a.gh) ? a.rel_y + ((int64_t)yi * a.gh + e) * a.dh
                                    : a.rel_x + ((int64_t)xi * a.gw + (e - a.gh)) * a.dh;
      float s = 0.f;
      for (int d = 0; d < a.dh; d += 4) {
        const float4 t = *reinterpret_cast<const float4*>(tab + d);
        s += qs[d] * t.x + qs[d + 1] * t.y + qs[d + 2] * t.z + qs[d + 3] * t.w;
      }
      if (e < a.gh) ry[e] = s; else rx[e - a.gh] = s;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }

  const float* prow = a.product + row * a.Nk;
  float mx = -INFINITY;
  for (int j = lane; j < a.Nk; j += 64) {
    float x = prow[j];
    if (rel) {
      const int ky = fast_div(j, inv_gw);
      // (x + ty) + tx: the reference adds the y term first (utils.py:159-168)
      x = (x + ry[ky]) + rx[j - ky * a.gw];
    }
    lg[j] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < a.Nk; j += 64) {
    const float e = expf(lg[j] - mx);
    lg[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();

  T* st = reinterpret_cast<T*>(a.a_state) + row * a.Nk;
  if (!a.gated) {
    for (int j = lane; j < a.Nk; j += 64) Store<T>::store(st + j, lg[j] / sum);
  } else {
    const int cnt = a.count ? a.count[b] : a.kcap;
    const int32_t* ix = a.idx + (int64_t)b * a.kcap;
    T* an = reinterpret_cast<T*>(a.a_new) + row * a.kcap;
    T* ad = reinterpret_cast<T*>(a.a_delta) + row * a.kcap;
    for (int jj = lane; jj < cnt; jj += 64) {
      const int j = ix[jj];
      const float v = Store<T>::round(lg[j] / sum);
      const float old = Store<T>::load(st + j);
      Store<T>::store(an + jj, v);
      Store<T>::store(ad + jj, v - old);
      Store<T>::store(st + j, v);
    }
  }
}

// =============================================================================================
// K6a: value delta gate, one thread per 4 channels of one (clip, token).
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void v_gate_kernel(const float* __restrict__ vsrc, int64_t v_rs,
                                                     const int32_t* __restrict__ idx,
                                                     const int32_t* __restrict__ count, int G, int N, int D, int kcap,
                                                     T* __restrict__ v_state, T* __restrict__ v_delta,
                                                     T* __restrict__ v_old, int gated, int transposed, int H,
                                                     const int32_t* tok_map, int groups_per_clip, int clip_rows,
                                                     const float* pad_row) {
  const int v4 = D >> 2;
  const int rows = gated ? kcap : N;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)G * rows * v4) return;
  const int c4 = (int)(e % v4);
  const int64_t r = e / v4;
  const int g = (int)(r / rows), ii = (int)(r - (int64_t)g * rows);
  int tok = ii;
  if (gated) {
    if (count != nullptr && ii >= count[g]) {
      if (transposed) {  // columns past count of the k-contiguous operands are defined (zero): the fused kernel
#pragma unroll           // stages whole 64-column chunks
        for (int q = 0; q < 4; ++q) {
          const int64_t o = ((int64_t)g * D + c4 * 4 + q) * kcap + ii;
          Store<T>::store(v_delta + o, 0.f);
          Store<T>::store(v_old + o, 0.f);
        }
      }
      return;
    }
    tok = idx[(int64_t)g * kcap + ii];
  }
  const float* row = evt_token_row(vsrc, v_rs, tok_map, groups_per_clip, clip_rows, pad_row, g, tok, N);
  const float4 v = *reinterpret_cast<const float4*>(row + c4 * 4);
  const float vv[4] = {v.x, v.y, v.z, v.w};
  T* st = v_state + ((int64_t)g * N + tok) * D + c4 * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float nv = Store<T>::round(vv[q]);
    if (gated) {
      const float old = Store<T>::load(st + q);
      const float dl = Store<T>::round(nv - old);
      // token-major (B,kcap,D) for evt_av, or k-contiguous (B,H,dh,kcap) for evt_softmax_av_gated
      const int64_t o = transposed ? ((int64_t)g * D + c4 * 4 + q) * kcap + ii
                                   : ((int64_t)g * kcap + ii) * D + c4 * 4 + q;
      Store<T>::store(v_delta + o, dl);
      Store<T>::store(v_old + o, nv - dl);  // v_n_tilde - v_delta_tilde, modules.py:294
    }
    Store<T>::store(st + q, nv);
  }
}

// K6a, transposed outputs through an LDS tile: a workgroup owns 64 selected tokens x 64 channels, reads
// v / v_state with whole-row-segment accesses, and writes v_delta^T / v_old^T (B, D, kcap) as 128-byte
// k-contiguous segments (the operand layout of evt_softmax_av_gated).
template <typename T>
__global__ __launch_bounds__(256) void v_gate_t_kernel(const float* __restrict__ vsrc, int64_t v_rs,
                                                       const int32_t* __restrict__ idx,
                                                       const int32_t* __restrict__ count, int N, int D, int kcap,
                                                       T* __restrict__ v_state, T* __restrict__ v_delta_t,
                                                       T* __restrict__ v_old_t) {
  __shared__ __attribute__((aligned(16))) unsigned char vg_smem[evt_vgate_lds<T>()];
  evt_v_gate_t_role<T>(vsrc, v_rs, idx, count, N, D, kcap, v_state, v_delta_t, v_old_t, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z,
                       vg_smem);   // evt_prep_roles.h
}

// =============================================================================================
// K6: pv[g, i, h*dh + d] (+)= round(sum_j A1[g,h,i,j] V1[g,j,h*dh+d]) (+ round(A2 . V2))
// 64 rows x dh columns per workgroup; each wave owns 16 rows x dh columns as dh/16 16x16x4 MFMA
// accumulators; K streamed through LDS in chunks of 32.
// =============================================================================================
struct AvArgs {
  const void* a1; const void* v1; const void* a2; const void* v2;
  int64_t lda; const int32_t* count; void* pv; float* out_f32;
  const int32_t* out_map; int groups_per_clip, clip_rows;
  int G, H, N, K, D, dh, gated;
};

template <typename T, int NT>  // NT = dh / 16
__global__ __launch_bounds__(256) void av_kernel(const AvArgs a) {
  constexpr int KT = 32, LDA_S = KT + 4;
  const int ldv = a.dh + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A1s = smem;                      // [64][LDA_S]
  float* A2s = A1s + 64 * LDA_S;
  float* V1s = A2s + 64 * LDA_S;          // [KT][ldv]
  float* V2s = V1s + KT * ldv;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gh_ = blockIdx.y, g = gh_ / a.H, h = gh_ - g * a.H;
  const int i0 = blockIdx.x * 64;
  const int b = a.out_map ? g / a.groups_per_clip : g;
  const int K = (a.gated && a.count) ? a.count[b] : a.K;
  const bool two = a.a2 != nullptr;

  const T* A1 = reinterpret_cast<const T*>(a.a1) + ((int64_t)g * a.H + h) * a.N * a.lda;
  const T* A2 = two ? reinterpret_cast<const T*>(a.a2) + ((int64_t)g * a.H + h) * a.N * a.lda : nullptr;
  const int64_t vrows = a.gated ? (a.lda) : a.K;  // rows per group in V1/V2: kcap (== lda) or Nk
  const T* V1 = reinterpret_cast<const T*>(a.v1) + (int64_t)g * vrows * a.D + h * a.dh;
  const T* V2 = two ? reinterpret_cast<const T*>(a.v2) + (int64_t)g * vrows * a.D + h * a.dh : nullptr;

  f32x4 acc1[NT], acc2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc1[t][r] = 0.f; acc2[t][r] = 0.f; }

  const int lr = lane & 15, lq = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += KT) {
    // stage A tiles: 64 rows x 32 k
    for (int e = tid; e < 64 * KT; e += 256) {
      const int r = e >> 5, kk = e & 31;
      const int i = i0 + r, k = k0 + kk;
      const bool ok = (i < a.N) && (k < K);
      A1s[r * LDA_S + kk] = ok ? Store<T>::load(A1 + (int64_t)i * a.lda + k) : 0.f;
      if (two) A2s[r * LDA_S + kk] = ok ? Store<T>::load(A2 + (int64_t)i * a.lda + k) : 0.f;
    }
    // stage V tiles: 32 k x dh
    for (int e = tid; e < KT * a.dh; e += 256) {
      const int kk = e / a.dh, d = e - kk * a.dh;
      const int k = k0 + kk;
      const bool ok = k < K;
      V1s[kk * ldv + d] = ok ? Store<T>::load(V1 + (int64_t)k * a.D + d) : 0.f;
      if (two) V2s[kk * ldv + d] = ok ? Store<T>::load(V2 + (int64_t)k * a.D + d) : 0.f;
    }
    __syncthreads();
    // lane quarter lq covers k in [8*lq, 8*lq+8) of this chunk (k-order is free inside a sum)
    const float* pa1 = A1s + (wave * 16 + lr) * LDA_S + lq * 8;
    const float* pa2 = A2s + (wave * 16 + lr) * LDA_S + lq * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float x1 = pa1[s];
      const float x2 = two ? pa2[s] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float y1 = V1s[(lq * 8 + s) * ldv + t * 16 + lr];
        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, y1, acc1[t], 0, 0, 0);
        if (two) {
          const float y2 = V2s[(lq * 8 + s) * ldv + t * 16 + lr];
          acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, y2, acc2[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: 16x16 C layout: col = lane & 15, row = (lane >> 4) * 4 + reg
  T* pv = reinterpret_cast<T*>(a.pv);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 16 + lq * 4 + r;
    if (i >= a.N) continue;
    int64_t orow = (int64_t)g * a.N + i;
    bool emit = true;
    if (a.out_map != nullptr) {
      const int tr = a.out_map[(int64_t)(g % a.groups_per_clip) * a.N + i];
      emit = tr >= 0;
      orow = (int64_t)(g / a.groups_per_clip) * a.clip_rows + tr;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = h * a.dh + t * 16 + lr;
      float v = Store<T>::round(acc1[t][r]);
      if (a.gated) {
        const float prev = Store<T>::load(pv + ((int64_t)g * a.N + i) * a.D + col);
        v = Store<T>::round(prev + v);                                   // product += a_n~ @ dv~
        if (two) v = Store<T>::round(v + Store<T>::round(acc2[t][r]));    // product += da~ @ v_old
      }
      if (pv != nullptr) Store<T>::store(pv + ((int64_t)g * a.N + i) * a.D + col, v);
      if (a.out_f32 != nullptr && emit) a.out_f32[orow * a.D + col] = v;
    }
  }
}

template <typename T>
int launch_av(const AvArgs& a, void* stream) {
  const int nt = (a.dh % 16) == 0 ? a.dh / 16 : 0;
  const size_t lds = (size_t)(2 * 64 * 36 + 2 * 32 * (a.dh + 4)) * sizeof(float);
  const dim3 grid((a.N + 63) / 64, a.G * a.H), block(256);
  if (grid.x == 0 || grid.y == 0) return EVT_OK;
  hipStream_t s = evt_stream(stream);
  switch (nt) {
    case 1: hipLaunchKernelGGL((av_kernel<T, 1>), grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL((av_kernel<T, 2>), grid, block, lds, s, a); break;
    case 3: hipLaunchKernelGGL((av_kernel<T, 3>), grid, block, lds, s, a); break;   // 48
    case 4: hipLaunchKernelGGL((av_kernel<T, 4>), grid, block, lds, s, a); break;
    case 5: hipLaunchKernelGGL((av_kernel<T, 5>), grid, block, lds, s, a); break;   // 80: ViT-H (1280 / 16)
    case 6: hipLaunchKernelGGL((av_kernel<T, 6>), grid, block, lds, s, a); break;   // 96
    case 7: hipLaunchKernelGGL((av_kernel<T, 7>), grid, block, lds, s, a); break;   // 112
    case 8: hipLaunchKernelGGL((av_kernel<T, 8>), grid, block, lds, s, a); break;
    default: return evt_fail(EVT_ERR_BAD_SHAPE, "evt_av: head dim %d must be a multiple of 16, <= 128", a.dh);
  }
  return evt_check_launch("evt_av");
}


// =============================================================================================
// K/V token pooling (Block._pool_tokens, blocks.py:303-326): average the k and v slices of the packed
// (B, qh*qw, 3D) token buffer over p0 x p1 cells of the token grid -> (B, kh*kw, 2D) = [k | v].
// One thread per 4 channels of one pooled token; window order (dy, dx) as ATen's avg_pool2d, then / area.
// =============================================================================================
__global__ __launch_bounds__(256) void pool_kv_kernel(const float* __restrict__ qkv, int B, int qw, int kh, int kw, int D,
                                                      int p0, int p1, float* __restrict__ kv) {
  const int v4 = (2 * D) >> 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * kh * kw * v4) return;
  const int c4 = (int)(e % v4);
  const int64_t cell = e / v4;
  const int kx = (int)(cell % kw), ky = (int)((cell / kw) % kh), b = (int)(cell / ((int64_t)kw * kh));
  const int N = kh * p0 * qw;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int dy = 0; dy < p0; ++dy)
    for (int dx = 0; dx < p1; ++dx) {
      const int tok = (ky * p0 + dy) * qw + kx * p1 + dx;
      const float4 t = *reinterpret_cast<const float4*>(qkv + ((int64_t)b * N + tok) * 3 * D + D + c4 * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  const float area = (float)(p0 * p1);
  *reinterpret_cast<float4*>(kv + cell * 2 * D + c4 * 4) = make_float4(s.x / area, s.y / area, s.z / area, s.w / area);
}

// Block._pool_index (blocks.py:525-540): map selected tokens to pooled cells, de-duplicate, ascending.
// One workgroup per clip: flags in LDS, then the same wavefront-ballot compaction as K1.
__global__ __launch_bounds__(256) void pool_index_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                                                         int kcap, int qw, int p0, int p1, int kw, int Nk, int kcap_k,
                                                         int32_t* __restrict__ idx_k, int32_t* __restrict__ count_k) {
  extern __shared__ __attribute__((aligned(16))) uint32_t pflag[];  // Nk flags + 4 wave sums
  uint32_t* wsum = pflag + Nk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int cnt = count ? count[b] : kcap;
  for (int j = tid; j < Nk; j += 256) pflag[j] = 0;
  __syncthreads();
  for (int i = tid; i < cnt; i += 256) {
    const int t = idx[(int64_t)b * kcap + i];
    const int y = t / qw, x = t - y * qw;
    pflag[(y / p0) * kw + x / p1] = 1;
  }
  __syncthreads();
  uint32_t run = 0;
  for (int base = 0; base < Nk; base += 256) {
    const int j = base + tid;
    const bool sel = j < Nk && pflag[j] != 0;
    const unsigned long long bal = __ballot(sel);
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t pos = run + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += wsum[w];
    if (sel && pos < (uint32_t)kcap_k) idx_k[(int64_t)b * kcap_k + pos] = j;
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) count_k[b] = (int32_t)run;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static int check_heads(const char* who, int D, int H, int* dh) {
  if (H <= 0 || D <= 0 || D % H != 0) return evt_fail(EVT_ERR_BAD_SHAPE, "%s: D=%d not divisible by H=%d", who, D, H);
  *dh = D / H;
  if ((*dh % 8) != 0 || *dh > 128) return evt_fail(EVT_ERR_BAD_SHAPE, "%s: head dim %d must be a multiple of 8, <= 128", who, *dh);
  return EVT_OK;
}

extern "C" int evt_qk(const evt_qk_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_qk: null descriptor");
  EVT_REQUIRE(d->q && d->k && d->product, EVT_ERR_BAD_ARG, "evt_qk: null q/k/product");
  EVT_REQUIRE(d->G >= 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0 && d->scale > 0.f, EVT_ERR_BAD_ARG, "evt_qk: bad sizes");
  EVT_REQUIRE(d->dh > 0 && (d->dh % 8) == 0 && d->dh <= 128, EVT_ERR_BAD_SHAPE,
              "evt_qk: head dim %d must be a multiple of 8, <= 128", d->dh);
  EVT_REQUIRE((d->q_rs & 3) == 0 && (d->q_hs & 3) == 0 && (d->q_bs & 3) == 0 && (d->k_rs & 3) == 0 &&
              (d->k_hs & 3) == 0 && (d->k_bs & 3) == 0, EVT_ERR_BAD_SHAPE, "evt_qk: strides must be multiples of 4 elements");
  EVT_REQUIRE(d->tok_map == nullptr || (d->groups_per_clip > 0 && d->pad_q && d->pad_k && !d->delta && d->Nq == d->Nk),
              EVT_ERR_BAD_ARG, "evt_qk: tok_map needs groups_per_clip, pad rows, full mode and Nq == Nk");
  if (d->delta) {
    EVT_REQUIRE(d->idx_q && d->idx_k && d->kcap_q >= 0 && d->kcap_k >= 0, EVT_ERR_BAD_ARG, "evt_qk: delta mode needs idx_q/idx_k");
    if (d->kcap_q == 0 && d->kcap_k == 0) return EVT_OK;
  }
  QkArgs a{d->q, d->q_bs, d->q_hs, d->q_rs, d->k, d->k_bs, d->k_hs, d->k_rs, d->product,
           d->idx_q, d->count_q, d->kcap_q, d->delta ? d->idx_q_rest : nullptr, d->idx_k, d->count_k, d->kcap_k,
           d->tok_map, d->tok_map ? d->groups_per_clip : 1, d->pad_q, d->pad_k,
           d->G, d->H, d->Nq, d->Nk, d->dh, d->scale, d->delta, d->split ? 1 : 0};
  return launch_qk(a, stream);
}

extern "C" int evt_softmax_gate(const evt_softmax_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_softmax_gate: null descriptor");
  EVT_REQUIRE(d->product && d->a_state, EVT_ERR_BAD_ARG, "evt_softmax_gate: null product/a_state");
  EVT_REQUIRE(d->B >= 0 && d->H > 0 && d->N > 0 && d->Nk > 0, EVT_ERR_BAD_ARG, "evt_softmax_gate: bad sizes");
  EVT_REQUIRE((d->rel_y == nullptr) == (d->rel_x == nullptr), EVT_ERR_BAD_ARG, "evt_softmax_gate: rel_y/rel_x must come together");
  int dh = 0;
  if (d->rel_y) {
    EVT_REQUIRE(d->qkv != nullptr, EVT_ERR_BAD_ARG, "evt_softmax_gate: rel-pos needs qkv");
    EVT_REQUIRE(d->gh > 0 && d->gw > 0 && d->gh * d->gw == d->Nk && d->qw > 0 && d->N % d->qw == 0, EVT_ERR_BAD_SHAPE,
                "evt_softmax_gate: rel-pos key grid %dx%d / query width %d do not match N=%d/Nk=%d", d->gh, d->gw, d->qw, d->N, d->Nk);
    int rc = check_heads("evt_softmax_gate", d->D, d->H, &dh);
    if (rc) return rc;
  }
  EVT_REQUIRE(d->tok_map == nullptr || (d->groups_per_clip > 0 && d->clip_rows > 0 && d->pad_row), EVT_ERR_BAD_ARG,
              "evt_softmax_gate: tok_map needs groups_per_clip, clip_rows and pad_row");
  if (d->gated) {
    EVT_REQUIRE(d->a_new && d->a_delta && d->idx && d->kcap >= 0, EVT_ERR_BAD_ARG, "evt_softmax_gate: gated mode needs a_new/a_delta/idx");
  }
  SmArgs a{d->product, d->qkv, d->rel_y, d->rel_x, d->tok_map, d->pad_row,
           d->tok_map ? d->groups_per_clip : 1, d->clip_rows,
           d->a_state, d->a_new, d->a_delta, d->idx, d->count,
           d->B, d->H, d->N, d->Nk, d->D, dh, d->kcap, d->rel_y ? d->gh : 0, d->rel_y ? d->gw : 0, d->rel_y ? d->qw : 1, d->gated};
  const int64_t rows = (int64_t)d->B * d->H * d->N;
  if (rows == 0) return EVT_OK;
  const size_t lds = (size_t)4 * (a.Nk + a.dh + a.gh + a.gw) * sizeof(float);
  EVT_REQUIRE(lds <= 160 * 1024, EVT_ERR_BAD_SHAPE, "evt_softmax_gate: row of %d keys does not fit LDS", d->Nk);
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = evt_stream(stream);
  EVT_DISPATCH_STORE(d->store, T, {
    EVT_ALLOW_LDS(softmax_gate_kernel<T>, lds);
    hipLaunchKernelGGL(softmax_gate_kernel<T>, grid, block, lds, s, a);
  });
  return evt_check_launch("evt_softmax_gate");
}

extern "C" int evt_v_gate(const float* v, int64_t v_rs, const int32_t* idx, const int32_t* count, int B, int N, int D, int kcap,
                          void* v_state, void* v_delta, void* v_old, int store, int gated, int transposed,
                          const int32_t* tok_map, int groups_per_clip, int clip_rows, const float* pad_row,
                          void* stream) {
  EVT_REQUIRE(v && v_state, EVT_ERR_BAD_ARG, "evt_v_gate: null pointer");
  EVT_REQUIRE(B >= 0 && N > 0 && D > 0 && (D & 3) == 0 && v_rs >= D && (v_rs & 3) == 0, EVT_ERR_BAD_ARG, "evt_v_gate: bad sizes");
  const float* qkv = v;
  if (gated) EVT_REQUIRE(idx && v_delta && v_old && kcap >= 0 && tok_map == nullptr, EVT_ERR_BAD_ARG, "evt_v_gate: gated mode needs idx/v_delta/v_old and no tok_map");
  EVT_REQUIRE(tok_map == nullptr || (groups_per_clip > 0 && clip_rows > 0 && pad_row), EVT_ERR_BAD_ARG,
              "evt_v_gate: tok_map needs groups_per_clip, clip_rows and pad_row");
  const int64_t n = (int64_t)B * (gated ? kcap : N) * (D / 4);
  if (n == 0) return EVT_OK;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = evt_stream(stream);
  if (gated && transposed && (D % 64) == 0 && (kcap % 8) == 0 && kcap > 0) {
    const dim3 tgrid((kcap + 63) / 64, D / 64, B);
    EVT_DISPATCH_STORE(store, T, {
      hipLaunchKernelGGL(v_gate_t_kernel<T>, tgrid, block, 0, s, qkv, v_rs, idx, count, N, D, kcap, (T*)v_state,
                         (T*)v_delta, (T*)v_old);
    });
    return evt_check_launch("evt_v_gate");
  }
  EVT_DISPATCH_STORE(store, T, {
    hipLaunchKernelGGL(v_gate_kernel<T>, grid, block, 0, s, qkv, v_rs, idx, count, B, N, D, kcap, (T*)v_state, (T*)v_delta,
                       (T*)v_old, gated, transposed, 0, tok_map, tok_map ? groups_per_clip : 1, clip_rows, pad_row);
  });
  return evt_check_launch("evt_v_gate");
}

extern "C" int evt_av(const evt_av_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_av: null descriptor");
  EVT_REQUIRE(d->a1 && d->v1, EVT_ERR_BAD_ARG, "evt_av: null a1/v1");
  EVT_REQUIRE((d->a2 == nullptr) == (d->v2 == nullptr), EVT_ERR_BAD_ARG, "evt_av: a2/v2 must come together");
  EVT_REQUIRE(d->pv || d->out_f32, EVT_ERR_BAD_ARG, "evt_av: no output");
  EVT_REQUIRE(!d->gated || d->pv, EVT_ERR_BAD_ARG, "evt_av: gated mode accumulates into pv");
  EVT_REQUIRE(d->B >= 0 && d->N > 0 && d->K >= 0 && d->lda >= d->K, EVT_ERR_BAD_ARG, "evt_av: bad sizes");
  EVT_REQUIRE(d->out_map == nullptr || (d->groups_per_clip > 0 && d->clip_rows > 0 && d->out_f32), EVT_ERR_BAD_ARG,
              "evt_av: out_map needs groups_per_clip, clip_rows and out_f32");
  int dh;
  int rc = check_heads("evt_av", d->D, d->H, &dh);
  if (rc) return rc;
  AvArgs a{d->a1, d->v1, d->a2, d->v2, d->lda, d->count, d->pv, d->out_f32, d->out_map,
           d->out_map ? d->groups_per_clip : 1, d->clip_rows, d->B, d->H, d->N, d->K, d->D, dh, d->gated};
  EVT_DISPATCH_STORE(d->store, T, { return launch_av<T>(a, stream); });
  return EVT_OK;
}

extern "C" int evt_pool_kv(const float* qkv, int B, int qh, int qw, int D, int p0, int p1, float* kv, void* stream) {
  EVT_REQUIRE(qkv && kv, EVT_ERR_BAD_ARG, "evt_pool_kv: null pointer");
  EVT_REQUIRE(B >= 0 && qh > 0 && qw > 0 && D > 0 && p0 > 0 && p1 > 0, EVT_ERR_BAD_ARG, "evt_pool_kv: bad sizes");
  EVT_REQUIRE(qh % p0 == 0 && qw % p1 == 0 && (D & 3) == 0, EVT_ERR_BAD_SHAPE,
              "evt_pool_kv: grid %dx%d must be divisible by the pool %dx%d (blocks.py:480-482) and D %% 4 == 0", qh, qw, p0, p1);
  const int kh = qh / p0, kw = qw / p1;
  const int64_t n = (int64_t)B * kh * kw * (2 * D / 4);
  if (n == 0) return EVT_OK;
  hipLaunchKernelGGL(pool_kv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, evt_stream(stream), qkv, B, qw, kh, kw,
                     D, p0, p1, kv);
  return evt_check_launch("evt_pool_kv");
}

extern "C" int evt_pool_index(const int32_t* idx, const int32_t* count, int B, int kcap, int qw, int p0, int p1, int kw,
                              int Nk, int kcap_k, int32_t* idx_k, int32_t* count_k, void* stream) {
  EVT_REQUIRE(idx && idx_k && count_k, EVT_ERR_BAD_ARG, "evt_pool_index: null pointer");
  EVT_REQUIRE(B >= 0 && kcap >= 0 && qw > 0 && p0 > 0 && p1 > 0 && kw > 0 && Nk > 0 && kcap_k > 0, EVT_ERR_BAD_ARG,
              "evt_pool_index: bad sizes");
  EVT_REQUIRE(Nk <= 32768, EVT_ERR_BAD_SHAPE, "evt_pool_index: Nk=%d exceeds 32768", Nk);
  if (B == 0) return EVT_OK;
  hipLaunchKernelGGL(pool_index_kernel, dim3(B), dim3(256), (size_t)(Nk + 4) * sizeof(uint32_t), evt_stream(stream), idx,
                     count, kcap, qw, p0, p1, kw, Nk, kcap_k, idx_k, count_k);
  return evt_check_launch("evt_pool_index");
}
