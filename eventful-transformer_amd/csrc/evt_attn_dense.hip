// evt_attn_dense.hip -- K8: whole (window or clip) attention in one launch for groups of <= 256 keys, head dim 64:
//
//   S = (q / scale) k^T                     Block._forward_attention, blocks.py:205-240 (windowed: 257-301, 346-376)
//   S += rel-pos terms                      utils.py:159-168
//   P = round(softmax(S));  out = round(P . round(v)), heads merged, un-windowed on write
//
// This is what the windowed EventfulTokenwiseBlocks of ViTDet run on EVERY frame (attention there is dense), what
// `Block` runs, and -- with the optional state outputs -- the first frame of an EventfulBlock clip.  One workgroup owns
// 32 query rows of one (group, head): K is streamed through LDS in 64-key chunks for S, the 32 x Nk score tile stays
// in LDS for the softmax (rows never leave the CU), V^T chunks reuse the K region for P.V.  Scores, probabilities and
// the per-head values are never written to HBM unless the caller asks for the states.
//
// Arithmetic is that of the unfused kernels (K4 / K5 / K6): S on the fp32-input MFMA (exact fp32 products) or, with qk_split,
// on the bf16 hi/lo split MFMA like evt_qk's split mode (three bf16 MFMAs per product, 16x the rate); expf,
// probabilities and values rounded to the store type T, P.V on the T-input MFMA (fp32: 32x32x2; bf16 / fp16: 32x32x16),
// result rounded to T.
#include "evt_attn_dense.h"
#include "evt_linear.h"   // split4 (fp32 -> bf16 hi / lo), bf16x8_t

#ifdef EVT_PROF   // phase timing of wave 0 of one workgroup (scripts/attn_prof.py --dense)
__device__ unsigned long long evt_prof_dense_buf[8];
#define DN_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define DN_TICK(slot) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

constexpr int AR = 32;     // query rows per workgroup
constexpr int DH = 64;     // head dim
constexpr int QP = DH + 4; // fp32 LDS pitch of the q / k tiles
constexpr int KC = 64;     // keys per staged chunk

// SPLIT: S = (q / scale) k^T and the rel-pos terms on v_mfma_f32_16x16x32_bf16 with q, k and the rel-pos tables as bf16
// hi + lo (three MFMAs per product, ~1e-5 relative -- the arithmetic of evt_qk / evt_softmax_av_gated / evt_attention_stream in
// split mode) instead of exact fp32 products on v_mfma_f32_16x16x4_f32 at 1/16 of the rate.  In-kernel phase timing of the
// ViTDet window launch (9 windows x 12 heads, 196 tokens, scripts/attn_prof.py --dense window) before: rel-pos dots 40 % of a
// workgroup's life (per-table-row VALU dots with DPP reductions), q.k^T 17 %.  With SPLIT the rel-pos terms are ONE more
// score product: the tile's q rows against the (<= 4 gh + qw gw) table rows it can meet as "virtual keys", table
// fragments loaded straight from L2, each product written to the row's (ky | kx) slot when the table row is that row's.

// P.V over keys [0, len) of the staged chunk (len % 16 == 0): a = P row of this lane, b = V^T row of this lane.
template <typename T> struct PvSweep;
template <> struct PvSweep<float> {
  static __device__ __forceinline__ f32x16 run(const float* a, const float* b, int len, int lh, f32x16 acc) {
    const int half = len >> 1;  // lane half lh covers k in [lh*half, lh*half + half), half % 8 == 0
    for (int q = 0; q < half; q += 4) {
      const float4 fa = *reinterpret_cast<const float4*>(a + lh * half + q);
      const float4 fb = *reinterpret_cast<const float4*>(b + lh * half + q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
    }
    return acc;
  }
};
template <> struct PvSweep<bf16_t> {
  static __device__ __forceinline__ f32x16 run(const bf16_t* a, const bf16_t* b, int len, int lh, f32x16 acc) {
    for (int kk = 0; kk < len; kk += 16) {
      const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(a + kk + 8 * lh);
      const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(b + kk + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
};
template <> struct PvSweep<f16_t> {
  static __device__ __forceinline__ f32x16 run(const f16_t* a, const f16_t* b, int len, int lh, f32x16 acc) {
    for (int kk = 0; kk < len; kk += 16) {
      const f16x8_t fa = *reinterpret_cast<const f16x8_t*>(a + kk + 8 * lh);
      const f16x8_t fb = *reinterpret_cast<const f16x8_t*>(b + kk + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    }
    return acc;
  }
};

__host__ __device__ inline int dense_sp(int N) { return ((N + 15) & ~15) + 4; }  // fp32 pitch of the score tile
__host__ __device__ inline int dense_tile_floats(int N) {                          // score tile; the q tile aliases it
  const int s = AR * dense_sp(N), q = AR * QP;
  return s > q ? s : q;
}
typedef float f32x4_acc __attribute__((ext_vector_type(4)));

// Three workgroups per CU (the 49 KB of LDS allow it): at most 168 registers per lane INCLUDING the 16 AGPRs of the P.V
// accumulator -- at 154 + 16 the kernel ran two, and the 756 workgroups of a 672^2 frame took two rounds (48.7 us; 504
// workgroups: 28.7 us).
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256, 3) void attn_dense_kernel(const DenseArgs a) {
  constexpr int TPF = 4 / (int)sizeof(T);       // T elements per float slot
  constexpr int VP = KC + 16 / (int)sizeof(T);  // V^T pitch in T elements (16-byte pad)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SP = dense_sp(a.N);
  float* Ks = reinterpret_cast<float*>(smem_raw);          // [KC][QP]  K chunk;  later V^T chunk [DH][VP] of T
  T* Vt = reinterpret_cast<T*>(Ks);
  float* red = Ks;                                         // [AR][DH]  epilogue
  float* Ss = Ks + KC * QP;                                // [AR][SP]  scores, then P (T, pitch SP * TPF)
  float* Qs = Ss;                                          // [AR][QP]  q rows until their MFMA fragments sit in registers
  T* Ps = reinterpret_cast<T*>(Ss);
  float* relv = Ss + dense_tile_floats(a.N);               // [AR][gh + gw]
  int* tmap = reinterpret_cast<int*>(relv + AR * (a.gh + a.gw));  // [N] clip row of each group token, -1 = padding

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware placement (dispatch is round-robin over the 8 XCDs in linear workgroup order, x fastest): all row tiles of
  // one (group, head) run on the same XCD, so its K and V rows are fetched into ONE private L2.
  int gh_ = blockIdx.y, tile_x = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const int p = blockIdx.x + gridDim.x * blockIdx.y, x = p & 7, s = p >> 3, hl = s / (int)gridDim.x;
    tile_x = s - hl * (int)gridDim.x;
    gh_ = hl * 8 + x;
  }
  const int g = gh_ / a.H, h = gh_ - g * a.H;
  const int i0 = tile_x * AR;
  const int64_t rs = 3 * (int64_t)a.D;
  const bool rel = a.rel_y != nullptr;
  const int nrel = a.gh + a.gw;

  // ---- window map of the group, once (every staging step below resolves rows through LDS, so its global loads
  //      are independent and all in flight together) ---------------------------------------------------------------
  for (int j = tid; j < a.N; j += 256)
    tmap[j] = a.tok_map ? a.tok_map[(int64_t)(g % a.groups_per_clip) * a.N + j] : j;
  __syncthreads();
  const float* clip = a.qkv + (int64_t)(g / a.groups_per_clip) * a.clip_rows * rs;
  auto row_ptr = [&](int j) -> const float* {   // j < N
    const int r = tmap[j];
    return r < 0 ? a.pad_row : clip + (int64_t)r * rs;
  };

  // Global loads are issued one phase ahead of their use (registers), so the chain of dependent memory latencies
  // per workgroup is: window map -> q + K chunk 0 -> tables; everything else overlaps MFMA / softmax work.
  constexpr int IT = KC * (DH / 4) / 256;   // float4 per thread per chunk
  auto load_k = [&](int c0, float4* k) {     // thread -> (key r, channels c4*4..+3): coalesced 256-byte rows
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + 256 * it, r = e / (DH / 4), c4 = e - r * (DH / 4), j = c0 + r;
      k[it] = (j < a.N) ? *reinterpret_cast<const float4*>(row_ptr(j) + a.D + h * DH + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_k = [&](const float4* k) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + 256 * it, r = e / (DH / 4), c4 = e - r * (DH / 4);
      *reinterpret_cast<float4*>(Ks + r * QP + c4 * 4) = k[it];
    }
  };
  auto load_v = [&](int c0, float4* v) {     // same coalesced mapping as K (a wave load touches 4 rows, not 64);
#pragma unroll                                // the transposing LDS writes below pay a 4-way bank conflict instead
    for (int it = 0; it < IT; ++it) {
      const int e = tid + 256 * it, jj = e / (DH / 4), c4 = e - jj * (DH / 4), j = c0 + jj;
      v[it] = (j < a.N) ? *reinterpret_cast<const float4*>(row_ptr(j) + 2 * a.D + h * DH + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_v = [&](const float4* v) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + 256 * it, jj = e / (DH / 4), c4 = e - jj * (DH / 4);
      Store<T>::store(Vt + (c4 * 4 + 0) * VP + jj, v[it].x);
      Store<T>::store(Vt + (c4 * 4 + 1) * VP + jj, v[it].y);
      Store<T>::store(Vt + (c4 * 4 + 2) * VP + jj, v[it].z);
      Store<T>::store(Vt + (c4 * 4 + 3) * VP + jj, v[it].w);
    }
  };

#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == 2 && blockIdx.y == gridDim.y / 2 && (threadIdx.x >> 6) == 0;
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif
  // ---- q rows + K chunk 0 ---------------------------------------------------------------------------------------
  float4 kr[IT];
  {
    float4 q[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + 256 * it, r = e / (DH / 4), c4 = e - r * (DH / 4), i = i0 + r;
      q[it] = (i < a.N) ? *reinterpret_cast<const float4*>(row_ptr(i) + h * DH + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    load_k(0, kr);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + 256 * it, r = e / (DH / 4), c4 = e - r * (DH / 4);
      *reinterpret_cast<float4*>(Qs + r * QP + c4 * 4) = q[it];
    }
    store_k(kr);
  }
  __syncthreads();
  DN_TICK(0);   // q rows + K chunk 0 staged
  // ---- q MFMA fragments -> registers, scaled: q / self.scale (blocks.py:514); a power-of-two scale makes the
  //      reciprocal multiply exact.  Exact mode, 16x16x4 tiles: lane = (row l15 of a 16-row half, k group kg of 16 channels).
  //      SPLIT, 16x16x32 bf16 tiles: k-block m holds channels 32 m + 8 kg .. + 8 as bf16 hi and lo.
  const int l15 = lane & 15, kg = lane >> 4;
  const float inv_scale = 1.0f / a.scale;
  const bool pow2 = (inv_scale * a.scale == 1.0f) && ((__float_as_uint(a.scale) & 0x007fffffu) == 0u);
  float4 qf[2][4];
  bf16x8_t qh[2][2], ql[2][2];
  {
#pragma unroll
    for (int hr = 0; hr < 2; ++hr)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int ch = SPLIT ? 32 * (m >> 1) + 8 * kg + 4 * (m & 1) : kg * 16 + 4 * m;
        float4 q = *reinterpret_cast<const float4*>(Qs + (hr * 16 + l15) * QP + ch);
        if (pow2) { q.x *= inv_scale; q.y *= inv_scale; q.z *= inv_scale; q.w *= inv_scale; }
        else { q.x /= a.scale; q.y /= a.scale; q.z /= a.scale; q.w /= a.scale; }
        qf[hr][m] = q;
      }
    if (SPLIT) {
#pragma unroll
      for (int hr = 0; hr < 2; ++hr)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          bf16x4_t h0, l0, h1, l1;
          split4(qf[hr][2 * m], &h0, &l0);
          split4(qf[hr][2 * m + 1], &h1, &l1);
          qh[hr][m] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          ql[hr][m] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
  }
  // (q / scale) . b for the 16 "keys" whose channel pieces (4 x float4 in this lane's fragment layout) are in kf
  auto scores16 = [&](const float4* kf, f32x4_acc* acc) __attribute__((always_inline)) {
    acc[0] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};
    if (!SPLIT) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int hr = 0; hr < 2; ++hr) {
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[hr][m].x, kf[m].x, acc[hr], 0, 0, 0);
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[hr][m].y, kf[m].y, acc[hr], 0, 0, 0);
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[hr][m].z, kf[m].z, acc[hr], 0, 0, 0);
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[hr][m].w, kf[m].w, acc[hr], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x4_t h0, l0, h1, l1;
        split4(kf[2 * m], &h0, &l0);
        split4(kf[2 * m + 1], &h1, &l1);
        const bf16x8_t kh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), kl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int hr = 0; hr < 2; ++hr) {
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[hr][m], kh, acc[hr], 0, 0, 0);
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[hr][m], kl, acc[hr], 0, 0, 0);
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[hr][m], kh, acc[hr], 0, 0, 0);
        }
      }
    }
  };
  auto frag_chan = [&](int m) __attribute__((always_inline)) { return SPLIT ? 32 * (m >> 1) + 8 * kg + 4 * (m & 1) : kg * 16 + 4 * m; };
  if (KC < a.N) load_k(KC, kr);   // K chunk 1 flies during the rel-pos terms and the first MFMA sweep
  if (rel && SPLIT && pow2) {
    // rel-pos terms (utils.py:159-168: UNSCALED q . table row) as one more score product: "virtual keys" = the table rows
    // rel_y[y][ky] of the <= 4 query-grid rows y this tile touches, then all rel_x[x][kx].  q / scale is a power-of-two
    // multiple of q, so scale * ((q / scale) . t) is the same fp32 value as q . t computed this way.  Wave w owns the
    // 16-row groups w, w + 4, ...; lane (l15, kg) loads its table row's channel pieces straight from L2.
    const int i_hi = min(i0 + AR, a.N) - 1;
    const int y_lo = i0 / a.qw, ny = i_hi / a.qw - y_lo + 1;
    const int ycnt = ny * a.gh, vtot = ycnt + a.qw * a.gw;
    const float* ytab = a.rel_y + (int64_t)y_lo * a.gh * DH;
    const float inv_gh = 1.0f / (float)a.gh, inv_gw_ = 1.0f / (float)a.gw, inv_qw = 1.0f / (float)a.qw;
    int ry_[2][4], rx_[2][4];   // grid coordinates of the lane's 8 rows (row 16 hr + 4 kg + r), -1 past the group
#pragma unroll
    for (int hr = 0; hr < 2; ++hr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * hr + 4 * kg + r;
        const int yi = fast_div(i, inv_qw);
        ry_[hr][r] = i < a.N ? yi : -1;
        rx_[hr][r] = i < a.N ? i - yi * a.qw : -1;
      }
    auto load_t = [&](int vg, float4* tf) __attribute__((always_inline)) {
      const int v = min(vg * 16 + l15, vtot - 1);
      const float* row = v < ycnt ? ytab + (int64_t)v * DH : a.rel_x + (int64_t)(v - ycnt) * DH;
#pragma unroll
      for (int m = 0; m < 4; ++m) tf[m] = *reinterpret_cast<const float4*>(row + frag_chan(m));
    };
    auto consume_t = [&](int vg, const float4* tf) __attribute__((always_inline)) {
      f32x4_acc acc[2];
      scores16(tf, acc);
      const int v = vg * 16 + l15;
      if (v < vtot) {
        const bool isy = v < ycnt;
        const int x = isy ? v : v - ycnt;
        const int sel = isy ? fast_div(x, inv_gh) : fast_div(x, inv_gw_);          // table's grid row offset / grid column
        const int e = isy ? x - sel * a.gh : a.gh + x - sel * a.gw;                // slot in the row's (ky | kx) list
#pragma unroll
        for (int hr = 0; hr < 2; ++hr)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool mine = isy ? (ry_[hr][r] == y_lo + sel) : (rx_[hr][r] == sel);
            if (mine) relv[(16 * hr + 4 * kg + r) * nrel + e] = acc[hr][r] * a.scale;
          }
      }
    };
    const int ngrp = (vtot + 15) / 16;
    float4 tA[4], tB[4];
    if (wave < ngrp) load_t(wave, tA);
    for (int vg = wave; vg < ngrp; vg += 8) {
      if (vg + 4 < ngrp) load_t(vg + 4, tB);
      consume_t(vg, tA);
      if (vg + 4 < ngrp) {
        if (vg + 8 < ngrp) load_t(vg + 8, tA);
        consume_t(vg + 4, tB);
      }
    }
  } else if (rel) {
    // rel-pos terms of the 32 rows (utils.py:159-168), per TABLE ROW: (yi, ky) for the <= 32/qw + 2 query grid rows
    // the tile touches, then (xi, kx) for every xi.  A 16-lane DPP row owns one 256-byte table row (one coalesced
    // 16-byte load per lane, fetched once per workgroup) and dots it with the raw q rows that use it -- those with
    // that yi / that xi -- reducing the 16 partial sums with row_shr DPP adds.
    const int i_hi = min(i0 + AR, a.N) - 1;
    const int y_lo = i0 / a.qw, ny = i_hi / a.qw - y_lo + 1;
    const int items = ny * a.gh + a.qw * a.gw;
    const int grp = tid >> 4, l16 = tid & 15;
    constexpr int RB = 8;   // table rows in flight per 16-lane group
    for (int w0 = grp; w0 < items; w0 += 16 * RB) {
      float4 t[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int w = w0 + 16 * u;
        t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w < items) {
          const float* tab;
          if (w < ny * a.gh) tab = a.rel_y + ((int64_t)y_lo * a.gh + w) * DH;        // (y_lo + w / gh, w % gh) is row y_lo*gh + w
          else tab = a.rel_x + (int64_t)(w - ny * a.gh) * DH;
          t[u] = *reinterpret_cast<const float4*>(tab + l16 * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int w = w0 + 16 * u;
        if (w >= items) break;   // uniform per 16-lane group (and the DPP row ops stay inside the group)
        const bool isy = w < ny * a.gh;
        int sel, e;
        if (isy) { sel = y_lo + w / a.gh; e = w - (sel - y_lo) * a.gh; }
        else { const int x = w - ny * a.gh; sel = x / a.gw; e = x - sel * a.gw + a.gh; }
        int i = isy ? max(i0, sel * a.qw) : i0 + (sel - i0 % a.qw + a.qw) % a.qw;
        const int i_end = isy ? min(i_hi, sel * a.qw + a.qw - 1) : i_hi;
        const int step = isy ? 1 : a.qw;
        for (; i <= i_end; i += 4 * step) {   // 4 independent q rows per pass: the LDS read -> FMA -> DPP chains overlap
          float s[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int iv = min(i + v * step, i_end);
            const float4 q = *reinterpret_cast<const float4*>(Qs + (iv - i0) * QP + l16 * 4);
            s[v] = (q.x * t[u].x + q.y * t[u].y) + (q.z * t[u].z + q.w * t[u].w);
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) s[v] = row16_sum_dpp(s[v]);
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (l16 == 15 && i + v * step <= i_end) relv[(i + v * step - i0) * nrel + e] = s[v];
        }
      }
    }
  }
  __syncthreads();   // rel-pos dots and fragment reads are done: the q tile's storage becomes the score tile
  DN_TICK(1);   // rel-pos dots, q fragments

  // ---- S = q k^T: 64-key chunks, wave w owns keys 16w .. 16w+15 of the chunk for all 32 rows -------------------------
  float4 vr[IT];
  for (int c0 = 0; c0 < a.N; c0 += KC) {
    if (c0 > 0) {
      store_k(kr);
      __syncthreads();
    }
    if (c0 + KC < a.N) { if (c0 > 0) load_k(c0 + KC, kr); }   // chunk 1 was issued above
    else load_v(0, vr);                                        // last K chunk: V chunk 0 flies from here through the softmax
    const int n0 = c0 + wave * 16;
    if (n0 < a.N) {  // wave-uniform
      f32x4_acc acc[2];
      const float* kb = Ks + (wave * 16 + l15) * QP;
      float4 kf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) kf[m] = *reinterpret_cast<const float4*>(kb + frag_chan(m));
      scores16(kf, acc);
      const int j = n0 + l15;
      if (j < a.N) {
#pragma unroll
        for (int hr = 0; hr < 2; ++hr)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = hr * 16 + 4 * kg + r;
            Ss[row * SP + j] = acc[hr][r];
            if (a.product != nullptr && i0 + row < a.N)
              a.product[((int64_t)gh_ * a.N + i0 + row) * a.N + j] = acc[hr][r];
          }
      }
    }
    __syncthreads();
  }
  DN_TICK(2);   // q.k^T chunks
  store_v(vr);   // the K region is free: V^T chunk 0 lands while the softmax runs (next barrier publishes both)

  // ---- softmax: wave w owns rows 8w .. 8w+7; P overwrites the row's scores in place (as T) ------------------------
  {
    const float inv_gw = rel ? 1.0f / (float)a.gw : 0.f;
    const int npad = (a.N + 15) & ~15;
    T* ast = reinterpret_cast<T*>(a.a_state);
    // All 8 rows of the wave together: their max / sum reductions are independent DPP chains that overlap, and the exps
    // use v_exp_f32 and one reciprocal per row like the fused gated kernel (in-kernel phase timing: the row-after-row
    // version with expf and a division per element took 2.3-2.8k ticks per row, a third of the workgroup's life).
    float x[8][4], mx[8], rinv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave * 8 + rr;
      const float* rv = relv + r * nrel;
      float m = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        x[rr][u] = -INFINITY;
        if (j < a.N) {
          float v = Ss[r * SP + j];
          if (rel) { const int ky = fast_div(j, inv_gw); v = (v + rv[ky]) + rv[a.gh + j - ky * a.gw]; }
          x[rr][u] = v;
        }
        m = fmaxf(m, x[rr][u]);
      }
      mx[rr] = m;
    }
    wave_max_dpp_rows<8>(mx);   // (one DPP instruction per step, the eight rows interleaved: evt_common.h)
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[rr][u] = (lane + 64 * u < a.N) ? __builtin_amdgcn_exp2f((x[rr][u] - mx[rr]) * 1.44269504088896340736f) : 0.f;
        sum += x[rr][u];
      }
      rinv[rr] = sum;
    }
    wave_sum_dpp_rows<8>(rinv);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) rinv[rr] = 1.0f / rinv[rr];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();   // every lane has read the fp32 rows before anyone overwrites them with T
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave * 8 + rr, i = i0 + r;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < npad) {
          const float p = (j < a.N && i < a.N) ? Store<T>::round(x[rr][u] * rinv[rr]) : 0.f;
          Store<T>::store(Ps + r * SP * TPF + j, p);
          if (ast != nullptr && j < a.N && i < a.N) Store<T>::store(ast + ((int64_t)gh_ * a.N + i) * a.N + j, p);
        }
      }
    }
  }
  __syncthreads();

  DN_TICK(3);   // softmax + state write
  // ---- P . V: V^T chunks of 64 keys in the K region; wave = (dh half, 32-key half of the chunk) ----------------------
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int dt = wave & 1, kh = wave >> 1;
  const int npad = (a.N + 15) & ~15;
  for (int c0 = 0; c0 < a.N; c0 += KC) {
    if (c0 > 0) {
      store_v(vr);
      __syncthreads();
    }
    if (c0 + KC < a.N) load_v(c0 + KC, vr);
    const int k0 = c0 + kh * (KC / 2);
    int len = npad - k0;
    len = len > KC / 2 ? KC / 2 : len;
    if (len > 0)
      acc = PvSweep<T>::run(Ps + lr * SP * TPF + k0, Vt + (dt * 32 + lr) * VP + kh * (KC / 2), len, lh, acc);
    __syncthreads();
  }

  DN_TICK(4);   // P.V chunks
  // ---- epilogue: add the two key halves, round, merge heads, un-window ------------------------------------------
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * lh) * DH + dt * 32 + lr] = acc[r];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* p = red + ((r & 3) + 8 * (r >> 2) + 4 * lh) * DH + dt * 32 + lr;
      *p = Store<T>::round(acc[r] + *p);
    }
  }
  __syncthreads();
  T* pv = reinterpret_cast<T*>(a.pv);
  for (int e = tid; e < AR * (DH / 4); e += 256) {
    const int r = e / (DH / 4), c4 = e - r * (DH / 4), i = i0 + r;
    if (i >= a.N) continue;
    const float4 v = *reinterpret_cast<const float4*>(red + r * DH + c4 * 4);
    if (pv != nullptr) {
      T* p = pv + ((int64_t)g * a.N + i) * a.D + h * DH + c4 * 4;
      Store<T>::store(p, v.x); Store<T>::store(p + 1, v.y); Store<T>::store(p + 2, v.z); Store<T>::store(p + 3, v.w);
    }
    if (a.out_f32 != nullptr) {
      const int tr = tmap[i];
      if (tr < 0) continue;  // padding token: dropped on un-windowing (blocks.py:346-376)
      const int64_t orow = (int64_t)(g / a.groups_per_clip) * a.clip_rows + tr;
      *reinterpret_cast<float4*>(a.out_f32 + orow * a.D + h * DH + c4 * 4) = v;
    }
  }
#ifdef EVT_PROF
  DN_TICK(5);   // epilogue
  if (prof_on && (threadIdx.x & 63) == 0)
    for (int q = 0; q < 8; ++q) evt_prof_dense_buf[q] = prof_acc[q];
#endif
}

template <typename T>
int launch_dense(const DenseArgs& a, int split, void* stream) {
  const size_t lds = ((size_t)KC * QP + (size_t)dense_tile_floats(a.N) + (size_t)AR * (a.gh + a.gw) + a.N) * sizeof(float);
  const dim3 grid((a.N + AR - 1) / AR, a.G * a.H);
  if (grid.y == 0) return EVT_OK;
  if (split) {
    EVT_ALLOW_LDS((attn_dense_kernel<T, true>), lds);
    hipLaunchKernelGGL((attn_dense_kernel<T, true>), grid, dim3(256), lds, evt_stream(stream), a);
  } else {
    EVT_ALLOW_LDS((attn_dense_kernel<T, false>), lds);
    hipLaunchKernelGGL((attn_dense_kernel<T, false>), grid, dim3(256), lds, evt_stream(stream), a);
  }
  return evt_check_launch("evt_attention_dense");
}

}  // namespace

extern "C" int evt_attention_dense(const evt_attn_dense_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_attention_dense: null descriptor");
  EVT_REQUIRE(d->qkv != nullptr && (d->out_f32 != nullptr || d->pv != nullptr), EVT_ERR_BAD_ARG, "evt_attention_dense: null qkv / no output");
  EVT_REQUIRE(d->G >= 0 && d->H > 0 && d->N > 0 && d->D == d->H * 64, EVT_ERR_BAD_SHAPE,
              "evt_attention_dense: head dim must be 64 (D=%d, H=%d); use evt_qk + evt_softmax_gate + evt_av", d->D, d->H);
  EVT_REQUIRE(d->N <= 256, EVT_ERR_BAD_SHAPE, "evt_attention_dense: %d keys per group, at most 256 fit the score tile; use evt_qk + evt_softmax_gate + evt_av", d->N);
  EVT_REQUIRE((d->rel_y == nullptr) == (d->rel_x == nullptr), EVT_ERR_BAD_ARG, "evt_attention_dense: rel_y/rel_x");
  EVT_REQUIRE(d->scale > 0.f, EVT_ERR_BAD_ARG, "evt_attention_dense: scale");
  if (d->rel_y) {
    EVT_REQUIRE(d->gh > 0 && d->gw > 0 && d->gh * d->gw == d->N && d->qw > 0 && d->N % d->qw == 0, EVT_ERR_BAD_SHAPE,
                "evt_attention_dense: rel-pos key grid %dx%d / query width %d do not match N=%d", d->gh, d->gw, d->qw, d->N);
  }
  if (d->tok_map) {
    EVT_REQUIRE(d->groups_per_clip > 0 && d->clip_rows > 0 && d->pad_row != nullptr && d->G % d->groups_per_clip == 0, EVT_ERR_BAD_ARG,
                "evt_attention_dense: window map needs groups_per_clip, clip_rows, pad_row");
    EVT_REQUIRE(d->pv == nullptr && d->a_state == nullptr && d->product == nullptr, EVT_ERR_BAD_ARG,
                "evt_attention_dense: state outputs are for un-windowed attention");
  }
  DenseArgs a{d->qkv, d->rel_y, d->rel_x, d->tok_map, d->pad_row, d->out_f32, d->product, d->a_state, d->pv,
              d->tok_map ? d->groups_per_clip : 1, d->tok_map ? d->clip_rows : d->N, d->G, d->H, d->N, d->D,
              d->rel_y ? d->gh : 0, d->rel_y ? d->gw : 0, d->rel_y ? d->qw : 1, d->scale, d->norm_ref, d->norm_parts};
  EVT_REQUIRE((d->norm_ref == nullptr) == (d->norm_parts == nullptr), EVT_ERR_BAD_ARG, "evt_attention_dense: norm_ref / norm_parts come together");
  // Resident form first (evt_attn_window.hip: the group's K / V staged once per (group, head)); the tiled kernel below keeps
  // the launches that want state outputs and the shapes whose planes do not fit a CU's LDS.
  if (a.G == 0) return EVT_OK;
  if (evt_launch_window(a, d->store, d->qk_split, evt_stream(stream))) return evt_check_launch("evt_attention_dense (resident)");
  EVT_REQUIRE(d->norm_ref == nullptr, EVT_ERR_BAD_SHAPE, "evt_attention_dense: norm_ref / norm_parts are outputs of the resident kernel only "
              "(no state outputs, planes within a CU's LDS: ask evt_attention_dense_resident first)");
  EVT_DISPATCH_STORE(d->store, T, { return launch_dense<T>(a, d->qk_split, stream); });
  return EVT_OK;
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof_dense(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_dense_buf), sizeof(unsigned long long) * 8);
}
#endif
