// evt_prep_roles.h -- the three launches that prepare a gated frame of evt_attention_stream (K9), as device functions: the
// rel-pos terms of every query token (evt_rel_terms, split arithmetic), the frame's key rows as bf16 hi / lo MFMA fragments in
// fragment-major order (the "key plane"), and the value delta gate with transposed outputs (evt_v_gate).  All three depend only
// on the updated token buffer (and the gate's index list), not on each other: their stand-alone kernels call these bodies, and
// evt_stream_prep runs them as roles of ONE launch (a role per workgroup range) -- two launches and their boundaries less per
// global block of a one-stream frame.  Block coordinates come in as arguments; `smem` is the role's LDS region.
#pragma once
#include "evt_linear.h"   // split4, bf16x8_t

namespace {

typedef float evt_f32x4_acc_t __attribute__((ext_vector_type(4)));
constexpr int EVT_SKH = 4;          // key plane: heads per workgroup
constexpr int EVT_VGATE_TP = 64 + 8;  // value gate: LDS pitch of its two 64 x 64 tiles

// LDS bytes of each role
__host__ __device__ inline size_t evt_rel_terms_lds(int qh, int qw, int gh, int gw) {
  return (size_t)((qw > qh ? qw : qh) + (gh > gw ? gh : gw)) * 68 * sizeof(float);
}
constexpr size_t EVT_KEY_PLANE_LDS = (size_t)EVT_SKH * 256 * 16;
template <typename T> constexpr size_t evt_vgate_lds() { return (size_t)2 * 64 * EVT_VGATE_TP * sizeof(T); }

// block (bx, by) of a (qh + qw) x (B H) grid
__device__ __forceinline__ void evt_rel_terms_mfma_role(const float* __restrict__ qkv, const float* __restrict__ rel_y,
                                                        const float* __restrict__ rel_x, int H, int N, int D, int gh, int gw,
                                                        int qw, float* __restrict__ terms, int bx, int by, float* rt_smem) {
  typedef evt_f32x4_acc_t f32x4_acc;
  constexpr int SKH = EVT_SKH; (void)SKH;
  constexpr int DH = 64, LP = DH + 4;
  const int qh = N / qw, nrel = gh + gw;
  const bool is_y = bx < qh;
  const int pos = is_y ? bx : bx - qh;   // y, or x
  const int nq = is_y ? qw : qh, nkey = is_y ? gh : gw;
  const int bh = by, b = bh / H, h = bh - b * H;
  float* qs = rt_smem;               // [nq][LP]
  float* ts = rt_smem + nq * LP;     // [nkey][LP]
  const float* tab = is_y ? rel_y + (int64_t)pos * gh * DH : rel_x + (int64_t)pos * gw * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  for (int e = tid; e < nq * (DH / 4); e += 256) {
    const int r = e / (DH / 4), c4 = e - r * (DH / 4);
    const int i = is_y ? pos * qw + r : r * qw + pos;
    *reinterpret_cast<float4*>(qs + r * LP + c4 * 4) =
        *reinterpret_cast<const float4*>(qkv + ((int64_t)b * N + i) * 3 * D + h * DH + c4 * 4);
  }
  for (int e = tid; e < nkey * (DH / 4); e += 256) {
    const int r = e / (DH / 4), c4 = e - r * (DH / 4);
    *reinterpret_cast<float4*>(ts + r * LP + c4 * 4) = *reinterpret_cast<const float4*>(tab + (int64_t)r * DH + c4 * 4);
  }
  __syncthreads();
  auto frag = [&](const float* rows, int n, int tile, int m, bf16x8_t* hi, bf16x8_t* lo) __attribute__((always_inline)) {
    const float* p = rows + min(tile * 16 + l15, n - 1) * LP + 32 * m + 8 * kg;   // k-block m: channels 32 m + 8 kg .. + 8
    bf16x4_t h0, l0, h1, l1;
    split4(*reinterpret_cast<const float4*>(p), &h0, &l0);
    split4(*reinterpret_cast<const float4*>(p + 4), &h1, &l1);
    *hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const int TQ = (nq + 15) >> 4, TK = (nkey + 15) >> 4;
  for (int tq = wave; tq < TQ; tq += 4) {
    bf16x8_t qhi[2], qlo[2];
    frag(qs, nq, tq, 0, &qhi[0], &qlo[0]);
    frag(qs, nq, tq, 1, &qhi[1], &qlo[1]);
    for (int tk = 0; tk < TK; ++tk) {
      f32x4_acc acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8_t thi, tlo;
        frag(ts, nkey, tk, m, &thi, &tlo);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qlo[m], thi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qhi[m], tlo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qhi[m], thi, acc, 0, 0, 0);
      }
      const int key = tk * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tq * 16 + 4 * kg + r;
        if (row < nq && key < nkey) {
          const int i = is_y ? pos * qw + row : row * qw + pos;
          terms[((int64_t)bh * N + i) * nrel + (is_y ? 0 : gh) + key] = acc[r];
        }
      }
    }
  }
}

// 16-key blocks of K9's key plane.  Without a rel-pos key grid: block kb = keys 16 kb .. 16 kb + 15.  With one (gw > 0) every
// grid ROW starts a new block -- block kb = keys (ky, 16 xb .. 16 xb + 15) with ky = kb / ceil(gw / 16), xb = kb % ceil(gw / 16),
// slots past the row's end are zero rows -- so that a block lies inside ONE grid row and its rel-pos terms are one row term
// (uniform per wave) + 16 consecutive column terms (evt_attn_stream.hip, pass A).
__host__ __device__ inline int evt_key_blocks(int N, int gh, int gw) { return gw > 0 ? gh * ((gw + 15) >> 4) : (N + 15) >> 4; }

// block (bx, by) of a (B NKB) x ceil(H / 4) grid, NKB = evt_key_blocks(N, gh, gw); `tile`: EVT_KEY_PLANE_LDS bytes
// Key rows: N rows per clip of `ksrc`, row stride k_rs floats, key channels at k_off (the packed token buffer: 3 D, D; the pooled
// (B,Nk,2D) buffer of evt_pool_kv: 2 D, 0).
__device__ __forceinline__ void evt_split_keys_role(const float* __restrict__ ksrc, int64_t k_rs, int k_off, uint4* __restrict__ out, int B, int H, int N,
                                                    int NKB, int gw, int bx, int by, uint4* tile) {
  constexpr int SKH = EVT_SKH, SDH = 64;
  const int b = bx / NKB, kb = bx - b * NKB;
  const int h0 = by * SKH, nh = min(SKH, H - h0);
  const int units = 16 * nh * 8;                     // (key i, head, 8-channel group c8), c8 fastest: 32 consecutive bytes each
  const int kbr = (gw + 15) >> 4, ky = gw > 0 ? kb / kbr : 0, kx0 = gw > 0 ? (kb - ky * kbr) * 16 : 0;
  for (int u = threadIdx.x; u < units; u += 256) {
    const int c8 = u & 7, hh = (u >> 3) % nh, i = (u >> 3) / nh;
    const int key = gw > 0 ? ky * gw + kx0 + i : kb * 16 + i;
    bf16x8_t hi = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
    if (key < N && (gw == 0 || kx0 + i < gw)) {
      const float* src = ksrc + ((int64_t)b * N + key) * k_rs + k_off + (h0 + hh) * SDH + c8 * 8;
      const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
      bf16x4_t hx, lx, hy, ly;
      split4(x, &hx, &lx);
      split4(y, &hy, &ly);
      hi = __builtin_shufflevector(hx, hy, 0, 1, 2, 3, 4, 5, 6, 7);
      lo = __builtin_shufflevector(lx, ly, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // final position p = hh * 256 + (2 m + hl) * 64 + kg * 16 + i; in LDS the low four bits are rotated by the piece's
    // (c8 + 8 hh): a wave's lanes share i and differ in (c8, hh), un-rotated they would all hit the same four banks
    const int m = c8 >> 2, kg = c8 & 3, rot = (i + c8 + 8 * hh) & 15;
    uint4* dst = tile + hh * 256 + (m * 2) * 64 + kg * 16 + rot;
    dst[0] = __builtin_bit_cast(uint4, hi);
    dst[64] = __builtin_bit_cast(uint4, lo);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nh * 256; e += 256) {
    const int hh = e >> 8, g = e >> 4, c8 = ((g >> 3) & 1) * 4 + (g & 3);
    out[(((int64_t)b * H + h0 + hh) * NKB + kb) * 256 + (e & 255)] = tile[(e & ~15) | ((e + c8 + 8 * hh) & 15)];
  }
}

// block (bx, by, bz) of a ceil(kcap / 64) x (D / 64) x B grid; `smem`: evt_vgate_lds<T>() bytes
template <typename T>
__device__ __forceinline__ void evt_v_gate_t_role(const float* __restrict__ vsrc, int64_t v_rs, const int32_t* __restrict__ idx,
                                                  const int32_t* __restrict__ count, int N, int D, int kcap, T* __restrict__ v_state,
                                                  T* __restrict__ v_delta_t, T* __restrict__ v_old_t, int bx, int by, int bz,
                                                  unsigned char* smem) {
  constexpr int TP = EVT_VGATE_TP;  // LDS pitch in elements (16-byte aligned rows)
  T* td = reinterpret_cast<T*>(smem);
  T* to = td + 64 * TP;
  const int tid = threadIdx.x;
  const int k0 = bx * 64, c0 = by * 64, b = bz;
  const int cnt = count ? count[b] : kcap;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = (tid >> 4) + 16 * it, c4 = (tid & 15) * 4, ii = k0 + r;
    float dl[4] = {0.f, 0.f, 0.f, 0.f}, vo[4] = {0.f, 0.f, 0.f, 0.f};
    if (ii < cnt) {
      const int tok = idx[(int64_t)b * kcap + ii];
      const float4 v = *reinterpret_cast<const float4*>(vsrc + ((int64_t)b * N + tok) * v_rs + c0 + c4);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      T* st = v_state + ((int64_t)b * N + tok) * D + c0 + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float nv = Store<T>::round(vv[q]);
        const float old = Store<T>::load(st + q);
        dl[q] = Store<T>::round(nv - old);
        vo[q] = Store<T>::round(nv - dl[q]);  // v_n_tilde - v_delta_tilde, modules.py:294
        Store<T>::store(st + q, nv);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Store<T>::store(td + (c4 + q) * TP + r, dl[q]);
      Store<T>::store(to + (c4 + q) * TP + r, vo[q]);
    }
  }
  __syncthreads();
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int e = tid; e < 64 * (64 / VEC); e += 256) {
    const int ch = e / (64 / VEC), kk = (e - ch * (64 / VEC)) * VEC;
    if (k0 + kk >= kcap) continue;  // kcap % VEC == 0 is required by the launcher
    const int64_t o = ((int64_t)b * D + c0 + ch) * kcap + k0 + kk;
    *reinterpret_cast<uint4*>(v_delta_t + o) = *reinterpret_cast<const uint4*>(td + ch * TP + kk);
    *reinterpret_cast<uint4*>(v_old_t + o) = *reinterpret_cast<const uint4*>(to + ch * TP + kk);
  }
}

}  // namespace
