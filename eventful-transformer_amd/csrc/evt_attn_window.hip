// evt_attn_window.hip -- K8, resident form: ONE workgroup per (group, head) -- or two, splitting its query rows -- with the
// group's keys and values staged ONCE into LDS and every wave running its own 32 query rows to completion:
//
//   S = (q / scale) k^T  (+ decomposed rel-pos terms)        Block._forward_attention, blocks.py:205-240 (windows: 257-301, 346-376)
//   P = round(softmax(S));  out = round(P . round(v)), heads merged, un-windowed on write                utils.py:159-168
//
// Why.  The tiled kernel (evt_attn_dense.hip: 32 query rows per workgroup) re-stages K and V for each of a window's 7 row
// tiles through five barrier-separated phases; a workgroup lives ~33 us for 32 x 196 scores, and the 756 / 2100 workgroups of a
// ViTDet 672^2 / 1024^2 window launch need 1-3 rounds of 768 slots: 39.9 / 101 us per launch at 27 TFLOP/s, 8 launches per frame.
// Here the (group, head)'s K rows (as bf16 hi | lo planes, key-major) and V rows (TRANSPOSED, [channel][key]) enter LDS once
// (<= 146 KB for 196 tokens with rel-pos terms: one workgroup per CU), and after that single barrier no wave waits for another:
//
//   * scores TRANSPOSED on v_mfma_f32_32x32x16_bf16: A = 32 keys (from LDS), B = the wave's 32 query rows (registers), so a
//     lane owns ONE query row (lane & 31) and 16 keys per 32-key block: the whole 32 x N score block stays in registers
//     (<= 128), row max / sum are in-lane reductions plus one exchange with lane ^ 32;
//   * the accumulator layout of S^T IS the B-operand layout of the next product: out^T[d][q] = sum_k V^T[d][k] P^T[k][q]
//     with the contraction index permuted the same way on both sides (a lane's registers 8t .. 8t + 7 of a block are keys
//     16t + 4 lh + {0..3, 8..11}) -- P never goes through LDS;
//   * rel-pos terms: only (query, table row) pairs that share the query's grid row / column are needed, so they are NOT virtual
//     keys of the main product (that would double it) but 16 x 16 tiles (v_mfma_f32_16x16x32_bf16) per query-grid row Y
//     (its qw queries against rel_y[Y][.]) and per column X (its qh queries against rel_x[X][.]): 28 small tiles for a 14 x 14
//     window, computed by all waves before the barrier into a [query][ky | kx] LDS table the softmax reads.
//
// Arithmetic (same modes as the tiled kernel).  split = 1: q, k and -- with an fp32 store type -- P, V as bf16 hi + lo, three
// MFMAs per product (~1e-5 relative, the arithmetic of the gated linears); with a 16-bit store type P and V are rounded to it
// exactly where the reference rounds (`_cast_matmul_2`, blocks.py:183-189) and the product is exact.  split = 0: fp32-input
// MFMA (32x32x2) for the scores and, for an fp32 store type, for P.V.  expf by v_exp_f32, one reciprocal per row.
#include "evt_attn_dense.h"
#include "evt_linear.h"   // split4, bf16x8_t

#ifdef EVT_PROF   // phase timing of wave 0 of one workgroup (scripts/attn_prof.py --dense window)
__device__ unsigned long long evt_prof_window_buf[12];
#define WN_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define WN_TICK(slot) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_acc __attribute__((ext_vector_type(4)));

constexpr int DH = 64;
constexpr int KPB = DH + 8;   // bf16 pitch of a K plane row (144 B: conflict-free ds_read_b128 over 16-lane groups)
constexpr int KPF = DH + 4;   // fp32 pitch of a K row in the exact mode
constexpr int MAXB = 8;       // 32-key blocks: N <= 256

struct WinLds {
  int NP, VP, TP;                       // padded key count, V^T row pitch (elements), terms row pitch (floats)
  size_t k_bytes, v_plane_bytes, v_bytes, terms_bytes, tmap_bytes, total;
};

__host__ __device__ inline WinLds win_lds(int N, int nrel, int store_bytes, bool split) {
  WinLds l;
  l.NP = (N + 31) & ~31;
  l.VP = l.NP + 4;                      // (NP + 4) / 2 words is odd: conflict-free 8-byte reads over a 32-lane half
  l.TP = nrel > 0 ? (nrel | 1) : 0;
  l.k_bytes = split ? (size_t)l.NP * KPB * 2 * 2 : (size_t)l.NP * KPF * 4;
  const int planes = (store_bytes == 4 && split) ? 2 : 1;
  const int esz = store_bytes == 4 ? (split ? 2 : 4) : 2;
  l.v_plane_bytes = (size_t)DH * l.VP * esz;
  l.v_bytes = l.v_plane_bytes * planes;
  l.terms_bytes = (size_t)N * l.TP * 4;
  l.tmap_bytes = (size_t)l.NP * 4;
  l.total = l.k_bytes + l.v_bytes + l.terms_bytes + l.tmap_bytes;
  return l;
}

template <typename T> struct Half;   // 16-bit store types: packed conversion of 4 fp32 values, and the MFMA
template <> struct Half<bf16_t> {
  typedef bf16x4_t v4; typedef bf16x8_t v8;
  static __device__ __forceinline__ v4 cvt(float a, float b, float c, float d) {
    bf16x4_t hi, lo;
    split4(make_float4(a, b, c, d), &hi, &lo);   // hi = rne_bf16
    return hi;
  }
  static __device__ __forceinline__ f32x16 mfma(v8 x, v8 y, f32x16 acc) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0); }
};
template <> struct Half<f16_t> {
  typedef f16x4_t v4; typedef f16x8_t v8;
  static __device__ __forceinline__ v4 cvt(float a, float b, float c, float d) { return (v4){(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d}; }
  static __device__ __forceinline__ f32x16 mfma(v8 x, v8 y, f32x16 acc) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0); }
};

template <typename T, bool SPLIT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_window_kernel(const DenseArgs a) {
  constexpr int NT = 64 * NW;
  constexpr bool F32 = sizeof(T) == 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WinLds L = win_lds(a.N, a.gh + a.gw, (int)sizeof(T), SPLIT);
  const int NP = L.NP, VP = L.VP, TP = L.TP, NB = NP >> 5;
  unsigned char* kreg = smem;
  unsigned char* vreg = smem + L.k_bytes;
  float* terms = reinterpret_cast<float*>(vreg + L.v_bytes);
  int* tmap = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(terms) + L.terms_bytes);
  __bf16* Khi = reinterpret_cast<__bf16*>(kreg);
  __bf16* Klo = Khi + (size_t)NP * KPB;
  float* Kf = reinterpret_cast<float*>(kreg);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int gh_ = blockIdx.y, g = gh_ / a.H, h = gh_ - g * a.H;
  const int64_t rs = 3 * (int64_t)a.D;
  const bool rel = a.rel_y != nullptr;

#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == 0 && blockIdx.y == gridDim.y / 2 && wave == 0;
  unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif
  // ---- window map of the group (every later row address resolves through LDS) ----------------------------------------
  for (int j = tid; j < NP; j += NT)
    tmap[j] = j < a.N ? (a.tok_map ? a.tok_map[(int64_t)(g % a.groups_per_clip) * a.N + j] : j) : -1;
  __syncthreads();
  const float* clip = a.qkv + (int64_t)(g / a.groups_per_clip) * a.clip_rows * rs;
  auto row_ptr = [&](int j) -> const float* {   // j < N
    const int r = tmap[j];
    return r < 0 ? a.pad_row : clip + (int64_t)r * rs;
  };

  // ---- this wave's 32 query rows, COALESCED (16 lanes x 16 bytes per row, four rows per instruction): they reach the MFMA fragment
  //      layout -- and the rel-pos items below, which need the q rows of ANY wave -- through one LDS block per wave inside the K plane
  //      region, which is written later.  (Read straight into B-operand fragments, an instruction touched 64 separate lines for
  //      16 useful bytes each, and the rel-pos items read every q row twice more the same way: round 6, as in evt_attn_gated.hip.)
  const int i0 = (blockIdx.x * NW + wave) * 32, iq = i0 + lr;
  const bool wave_on = i0 < a.N, q_on = iq < a.N;
  constexpr int QP = DH + 4;   // row pitch (floats) of a q block: the fragment reads of 16 consecutive rows cover the 64 banks once
  float* const qblocks = reinterpret_cast<float*>(kreg);
  f32x4 qg[8];
  if (wave_on) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i0 + (lane >> 4) + 4 * i;
      qg[i] = r < a.N ? *reinterpret_cast<const f32x4*>(row_ptr(r) + h * DH + 4 * (lane & 15)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  WN_TICK(0);   // window map + q row requests
  // ---- staging.  Every global load of the phase is requested before the first one is consumed: the first batch of rel-pos
  //      items, the K rows, the V rows -- a workgroup then waits for ~one round trip, not one per step (in-kernel phase timing
  //      of the first version, which loaded and consumed step by step: 23k of a workgroup's 61k ticks here). -----------------
  // rel-pos terms (utils.py:159-168, UNSCALED q . table row) as 16 x 16 tiles: item = (grid row Y | grid column X, 16 of
  // its queries, 16 of its table rows); the waves deal the items round robin, BS in flight per wave.
  constexpr int BS = 2;   // (4 in flight for the 4-wave shape spilled: 64 + 64 staging registers of K / V on top)
  const int qh_ = rel ? a.N / a.qw : 1;
  const int l15 = lane & 15, kg = lane >> 4;
  const int nqy = (a.qw + 15) >> 4, nty = (a.gh + 15) >> 4, nqx = (qh_ + 15) >> 4, ntx = (a.gw + 15) >> 4;
  const int items_y = rel ? qh_ * nqy * nty : 0, items = rel ? items_y + a.qw * nqx * ntx : 0;
  auto frag_chan = [&](int m) __attribute__((always_inline)) { return SPLIT ? 32 * (m >> 1) + 8 * kg + 4 * (m & 1) : 16 * kg + 4 * m; };
  f32x4 tf[2 * BS][4], qf[BS][4];   // table rows of two batches (the next one is in flight while this one is consumed); q fragments from LDS
  int it_qi[2 * BS], it_e[2 * BS], it_q[2 * BS];   // lane's query token (-1: none), lane's table slot in the terms row (-1: none), the token whose q row it reads
  auto load_item = [&](int it, int u) __attribute__((always_inline)) {
    const bool isy = it < items_y;
    const int x = isy ? it : it - items_y;
    const int nq = isy ? nqy : nqx, nt = isy ? nty : ntx;
    const int sel = x / (nq * nt), rem = x - sel * (nq * nt), qc = rem / nt, tc = rem - qc * nt;
    const int qn = isy ? a.qw : qh_, tn = isy ? a.gh : a.gw;
    const int ql = qc * 16 + l15, tl = tc * 16 + l15;
    const int qtok = isy ? sel * a.qw + ql : ql * a.qw + sel;
    it_qi[u] = ql < qn ? qtok : -1;
    it_e[u] = tl < tn ? (isy ? tl : a.gh + tl) : -1;
    const float* trow_ = (isy ? a.rel_y + ((int64_t)sel * a.gh + min(tl, tn - 1)) * DH : a.rel_x + ((int64_t)sel * a.gw + min(tl, tn - 1)) * DH);
    it_q[u] = ql < qn ? qtok : 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) tf[u][m] = *reinterpret_cast<const f32x4*>(trow_ + frag_chan(m));
  };
  auto item_q = [&](int u) __attribute__((always_inline)) {   // the item's query fragments from the q blocks (token t: block t / 32, row t % 32)
    const float* qrow_ = qblocks + (size_t)it_q[u] * QP;   // (one workgroup per (group, head): every token's block is here)
#pragma unroll
    for (int m = 0; m < 4; ++m) qf[u % BS][m] = *reinterpret_cast<const f32x4*>(qrow_ + frag_chan(m));
  };
  auto consume_item = [&](int u) __attribute__((always_inline)) {
    // D[i = table row 4 kg' + r][j = query l15]: lane (l15, kg) holds rows 4 kg + r of ITS query column
    item_q(u);
    f32x4_acc acc = {0.f, 0.f, 0.f, 0.f};
    if (SPLIT) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x4_t th0, tl0, th1, tl1, qh0, ql0, qh1, ql1;
        split4(make_float4(tf[u][2 * s2][0], tf[u][2 * s2][1], tf[u][2 * s2][2], tf[u][2 * s2][3]), &th0, &tl0);
        split4(make_float4(tf[u][2 * s2 + 1][0], tf[u][2 * s2 + 1][1], tf[u][2 * s2 + 1][2], tf[u][2 * s2 + 1][3]), &th1, &tl1);
        split4(make_float4(qf[u % BS][2 * s2][0], qf[u % BS][2 * s2][1], qf[u % BS][2 * s2][2], qf[u % BS][2 * s2][3]), &qh0, &ql0);
        split4(make_float4(qf[u % BS][2 * s2 + 1][0], qf[u % BS][2 * s2 + 1][1], qf[u % BS][2 * s2 + 1][2], qf[u % BS][2 * s2 + 1][3]), &qh1, &ql1);
        const bf16x8_t th = __builtin_shufflevector(th0, th1, 0, 1, 2, 3, 4, 5, 6, 7), tl = __builtin_shufflevector(tl0, tl1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8_t qh8 = __builtin_shufflevector(qh0, qh1, 0, 1, 2, 3, 4, 5, 6, 7), ql8 = __builtin_shufflevector(ql0, ql1, 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl, qh8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, ql8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, qh8, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf[u][m][c], qf[u % BS][m][c], acc, 0, 0, 0);
    }
    // the table slot of row 4 kg + r is the `e` of lane (4 kg + r): fetch it from that lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = __shfl(it_e[u], 4 * kg + r, 64);
      if (it_qi[u] >= 0 && e >= 0) terms[it_qi[u] * TP + e] = acc[r];
    }
  };
  auto load_batch = [&](int n, int par) __attribute__((always_inline)) {   // items wave + NW (BS n + u) -> register set `par`
#pragma unroll
    for (int u = 0; u < BS; ++u) {
      const int it = wave + NW * (BS * n + u);
      if (it < items) load_item(it, par * BS + u);   // wave-uniform
    }
  };
  auto consume_batch = [&](int n, int par) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < BS; ++u) {
      const int it = wave + NW * (BS * n + u);
      if (it < items) consume_item(par * BS + u);
    }
  };
  const int batches = rel ? (items - wave + NW * BS - 1) / (NW * BS) : 0;   // of this wave
  if (batches > 0) load_batch(0, 0);
  // K rows (key-major; split mode: bf16 hi | lo planes) and V rows (transposed: [channel][key], four consecutive keys of one
  // channel per store)
  constexpr int KIT = (MAXB * 32 * 16 + NT - 1) / NT, VIT = (MAXB * 8 * 16 + NT - 1) / NT;
  f32x4 kr[KIT], vv[VIT][4];
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int e = tid + NT * it, j = e >> 4, c4 = e & 15;
    kr[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (j < a.N) kr[it] = *reinterpret_cast<const f32x4*>(row_ptr(j) + a.D + h * DH + c4 * 4);
  }
#pragma unroll
  for (int it = 0; it < VIT; ++it) {
    const int e = tid + NT * it, jq = e >> 4, c4 = e & 15;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = 4 * jq + u;
      vv[it][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (j < a.N) vv[it][u] = *reinterpret_cast<const f32x4*>(row_ptr(j) + 2 * a.D + h * DH + c4 * 4);
    }
  }
  WN_TICK(1);   // requests issued
  // q rows -> the wave's block -> this lane's B-operand fragments (SPLIT: channels 16 s + 8 lh + {0..3, 4..7}, s = piece / 2;
  // exact: channels 8 m + 4 lh + 0..3, m = piece)
  float4 qraw[8];
  if (wave_on) {
    float* qst = qblocks + (size_t)wave * 32 * QP;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(qst + ((lane >> 4) + 4 * i) * QP + 4 * (lane & 15)) = qg[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int ch = SPLIT ? 16 * (m >> 1) + 8 * lh + 4 * (m & 1) : 8 * m + 4 * lh;
      const f32x4 v = *reinterpret_cast<const f32x4*>(qst + lr * QP + ch);
      qraw[m] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();   // every q block is staged (the rel-pos items read any of them)
  // rel-pos items: table rows from global (a batch ahead), q fragments from the blocks
  for (int n = 0; n < batches; n += 2) {   // (two batches per trip: the register set of a batch is a compile-time index)
    if (n + 1 < batches) load_batch(n + 1, 1);
    consume_batch(n, 0);
    if (n + 1 < batches) {
      if (n + 2 < batches) load_batch(n + 2, 0);
      consume_batch(n + 1, 1);
    }
  }
  WN_TICK(3);   // rel-pos items
  if (rel) __syncthreads();   // (launch-uniform) every wave is done with the q blocks: the K planes overwrite them
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int e = tid + NT * it, j = e >> 4, c4 = e & 15;
    if (j < NP) {
      if (SPLIT) {
        bf16x4_t hi, lo;
        split4(make_float4(kr[it][0], kr[it][1], kr[it][2], kr[it][3]), &hi, &lo);
        *reinterpret_cast<bf16x4_t*>(Khi + (size_t)j * KPB + c4 * 4) = hi;
        *reinterpret_cast<bf16x4_t*>(Klo + (size_t)j * KPB + c4 * 4) = lo;
      } else {
        *reinterpret_cast<f32x4*>(Kf + (size_t)j * KPF + c4 * 4) = kr[it];
      }
    }
  }
#pragma unroll
  for (int it = 0; it < VIT; ++it) {
    const int e = tid + NT * it, jq = e >> 4, c4 = e & 15;
    if (4 * jq < NP) {   // uniform per 16-lane group
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x0 = vv[it][0][c], x1 = vv[it][1][c], x2 = vv[it][2][c], x3 = vv[it][3][c];
        const size_t off = (size_t)(4 * c4 + c) * VP + 4 * jq;
        if constexpr (F32 && SPLIT) {
          bf16x4_t hi, lo;
          split4(make_float4(x0, x1, x2, x3), &hi, &lo);
          *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(vreg) + off) = hi;
          *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(vreg + L.v_plane_bytes) + off) = lo;
        } else if constexpr (F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(vreg) + off) = (f32x4){x0, x1, x2, x3};
        } else {
          *reinterpret_cast<typename Half<T>::v4*>(reinterpret_cast<uint16_t*>(vreg) + off) = Half<T>::cvt(x0, x1, x2, x3);
        }
      }
    }
  }
  WN_TICK(2);   // K / V planes written
  __syncthreads();   // K, V^T and the rel-pos terms are resident: behind this barrier no wave waits for another
  WN_TICK(4);   // barrier
  if (!wave_on) return;

  // ---- q / self.scale (blocks.py:514): a power-of-two scale makes the reciprocal multiply exact -------------------------
  const float inv_scale = 1.0f / a.scale;
  const bool pow2 = (inv_scale * a.scale == 1.0f) && ((__float_as_uint(a.scale) & 0x007fffffu) == 0u);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    float4 q = qraw[m];
    if (pow2) { q.x *= inv_scale; q.y *= inv_scale; q.z *= inv_scale; q.w *= inv_scale; }
    else { q.x /= a.scale; q.y /= a.scale; q.z /= a.scale; q.w /= a.scale; }
    qraw[m] = q;
  }
  bf16x8_t qh[4], ql[4];
  if (SPLIT) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x4_t h0, l0, h1, l1;
      split4(qraw[2 * s], &h0, &l0);
      split4(qraw[2 * s + 1], &h1, &l1);
      qh[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      ql[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }

  // ---- S^T = K (q / scale)^T: block b = keys 32 b .. 32 b + 31; lane (lr, lh), register r = 4 g + e holds
  //      key 32 b + 8 g + 4 lh + e of query row lr ---------------------------------------------------------------------
  // Rel-pos terms in split mode (key grids up to 16 x 16): S^T[key][q] += sum_c E[key][c] T[c][q] with T = the query's
  // (ky | kx) terms (bf16 hi + lo from the LDS table, this lane's B fragments for all blocks) and E the ONE-HOT rows
  // "c == ky(key)" / "c == 16 + kx(key)" built in registers (exact in bf16): two more k-steps of the score product instead of
  // two LDS reads, an index division and two adds per score on the VALU (10k of a workgroup's 51k ticks).
  const float inv_gw = rel ? 1.0f / (float)a.gw : 0.f;
  const float* trow = terms + (q_on ? iq : 0) * TP;
  const bool onehot = SPLIT && rel && a.gh <= 16 && a.gw <= 16;
  bf16x8_t tyh, tyl, txh, txl;
  if (onehot) {
    float ty[8], tx[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int c = 8 * lh + n;
      ty[n] = trow[min(c, a.gh - 1)];
      tx[n] = trow[a.gh + min(c, a.gw - 1)];
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int c = 8 * lh + n;
      ty[n] = c < a.gh ? ty[n] : 0.f;
      tx[n] = c < a.gw ? tx[n] : 0.f;
    }
    bf16x4_t h0, l0, h1, l1;
    split4(make_float4(ty[0], ty[1], ty[2], ty[3]), &h0, &l0);
    split4(make_float4(ty[4], ty[5], ty[6], ty[7]), &h1, &l1);
    tyh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    tyl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    split4(make_float4(tx[0], tx[1], tx[2], tx[3]), &h0, &l0);
    split4(make_float4(tx[4], tx[5], tx[6], tx[7]), &h1, &l1);
    txh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    txl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  auto one_hot8 = [&](int pos) __attribute__((always_inline)) {   // bf16 1.0 at element pos (0..7), zeros elsewhere / out of range
    const uint32_t v = (pos & 1) ? 0x3F800000u : 0x00003F80u;
    const int w = pos >> 1;   // negative or > 3: no dword matches
    union { uint4 u; bf16x8_t b; } c;
    c.u = make_uint4(w == 0 ? v : 0u, w == 1 ? v : 0u, w == 2 ? v : 0u, w == 3 ? v : 0u);
    return c.b;
  };
  f32x16 S[MAXB];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) S[b][r] = 0.f;
    if (b < NB) {
      if (onehot) {   // wave-uniform
        const int key = 32 * b + lr;                       // A operand row of this lane
        const int ky = fast_div(key, inv_gw), kx = key - ky * a.gw;
        const bool live = key < a.N;
        const bf16x8_t ey = one_hot8(live ? ky - 8 * lh : -2), ex = one_hot8(live ? kx - 8 * lh : -2);
        S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ey, tyl, S[b], 0, 0, 0);
        S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ex, txl, S[b], 0, 0, 0);
        S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ey, tyh, S[b], 0, 0, 0);
        S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ex, txh, S[b], 0, 0, 0);
      }
      if (SPLIT) {
        const __bf16* kh_row = Khi + (size_t)(32 * b + lr) * KPB + 8 * lh;
        const __bf16* kl_row = Klo + (size_t)(32 * b + lr) * KPB + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(kh_row + 16 * s);
          const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(kl_row + 16 * s);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], S[b], 0, 0, 0);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], S[b], 0, 0, 0);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], S[b], 0, 0, 0);
        }
      } else {
        const float* k_row = Kf + (size_t)(32 * b + lr) * KPF + 4 * lh;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float4 kf = *reinterpret_cast<const float4*>(k_row + 8 * m);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qraw[m].x, S[b], 0, 0, 0);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qraw[m].y, S[b], 0, 0, 0);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qraw[m].z, S[b], 0, 0, 0);
          S[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qraw[m].w, S[b], 0, 0, 0);
        }
      }
    }
  }

  WN_TICK(5);   // q fragments + S^T products
  // ---- softmax of the lane's row (its other half sits in lane ^ 32) -------------------------------------------------------
  float mx = -INFINITY;
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
    if (b < NB) {
      if (rel && !onehot) {
        // (attn + rel_h) + rel_w, utils.py:166-168.  The 32 table reads of a block go out together through clamped, branch-free
        // addresses (first version: a predicated read + wait per score, 194 ticks each -- 36 % of the workgroup's life).
        float th[16], tw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = min(32 * b + 8 * (r >> 2) + 4 * lh + (r & 3), a.N - 1);
          const int ky = fast_div(key, inv_gw), kx = key - ky * a.gw;
          th[r] = trow[ky];
          tw[r] = trow[a.gh + kx];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) S[b][r] = (S[b][r] + th[r]) + tw[r];
      }
      if (32 * b + 32 > a.N) {   // only the last block has keys past N (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * b + 8 * (r >> 2) + 4 * lh + (r & 3);
          S[b][r] = key < a.N ? S[b][r] : -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[b][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  WN_TICK(6);   // rel-pos adds, masks, row max
  float sum = 0.f;
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
    if (b < NB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f((S[b][r] - mx) * 1.44269504088896340736f);   // exp2(-inf) = 0 for masked keys
        S[b][r] = p;
        sum += p;
      }
    }
  sum += __shfl_xor(sum, 32, 64);
  const float rinv = 1.0f / sum;
  WN_TICK(7);   // exps, row sum

  // ---- out^T = V^T P^T: MFMA step (b, t) contracts keys 32 b + 16 t + 4 lh' + {0..3, 8..11} -- exactly the keys a lane's
  //      registers 8 t .. 8 t + 7 of block b hold -- against the V^T row pieces at the same keys ------------------------
  f32x16 O[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
    if (b < NB) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float p[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) p[n] = Store<T>::round(S[b][8 * t + n] * rinv);
        const int k0 = 32 * b + 16 * t + 4 * lh;
        if constexpr (F32 && SPLIT) {
          bf16x4_t h0, l0, h1, l1;
          split4(make_float4(p[0], p[1], p[2], p[3]), &h0, &l0);
          split4(make_float4(p[4], p[5], p[6], p[7]), &h1, &l1);
          const bf16x8_t ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          const __bf16* vh_base = reinterpret_cast<const __bf16*>(vreg);
          const __bf16* vl_base = reinterpret_cast<const __bf16*>(vreg + L.v_plane_bytes);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const size_t off = (size_t)(32 * d + lr) * VP + k0;
            const bf16x4_t vh0 = *reinterpret_cast<const bf16x4_t*>(vh_base + off), vh1 = *reinterpret_cast<const bf16x4_t*>(vh_base + off + 8);
            const bf16x4_t vl0 = *reinterpret_cast<const bf16x4_t*>(vl_base + off), vl1 = *reinterpret_cast<const bf16x4_t*>(vl_base + off + 8);
            const bf16x8_t vh = __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7), vl = __builtin_shufflevector(vl0, vl1, 0, 1, 2, 3, 4, 5, 6, 7);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, O[d], 0, 0, 0);
          }
        } else if constexpr (F32) {
          const float* vf_base = reinterpret_cast<const float*>(vreg);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const size_t off = (size_t)(32 * d + lr) * VP + k0;
            const float4 v0 = *reinterpret_cast<const float4*>(vf_base + off), v1 = *reinterpret_cast<const float4*>(vf_base + off + 8);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, p[0], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, p[1], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.z, p[2], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.w, p[3], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, p[4], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, p[5], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.z, p[6], O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.w, p[7], O[d], 0, 0, 0);
          }
        } else {
          typedef typename Half<T>::v4 h4; typedef typename Half<T>::v8 h8;
          const h4 p0 = Half<T>::cvt(p[0], p[1], p[2], p[3]), p1 = Half<T>::cvt(p[4], p[5], p[6], p[7]);   // exact: p is already rounded
          const h8 pp = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
          const uint16_t* v_base = reinterpret_cast<const uint16_t*>(vreg);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const size_t off = (size_t)(32 * d + lr) * VP + k0;
            const h4 v0 = *reinterpret_cast<const h4*>(v_base + off), v1 = *reinterpret_cast<const h4*>(v_base + off + 8);
            O[d] = Half<T>::mfma(__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), pp, O[d]);
          }
        }
      }
    }

  WN_TICK(8);   // P conversion + P.V products
  // ---- epilogue: round, merge heads, un-window: lane (query lr, lh), register 4 g + e = channel 32 d + 8 g + 4 lh + e ----
  const int tr = q_on ? tmap[iq] : -1;   // < 0: padding token, dropped on un-windowing (blocks.py:346-376)
  float ss = 0.f;   // norm_ref: this lane's 32 channels of || out - ref ||^2 (the projection gate's delta norm, per head)
  if (tr >= 0) {
  float* orow = a.out_f32 + ((int64_t)(g / a.groups_per_clip) * a.clip_rows + tr) * a.D + h * DH + 4 * lh;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      float4 v;
      v.x = Store<T>::round(O[d][4 * gq + 0]);
      v.y = Store<T>::round(O[d][4 * gq + 1]);
      v.z = Store<T>::round(O[d][4 * gq + 2]);
      v.w = Store<T>::round(O[d][4 * gq + 3]);
      *reinterpret_cast<float4*>(orow + 32 * d + 8 * gq) = v;
      if (a.norm_ref != nullptr) {   // launch-uniform
        const float4 p = *reinterpret_cast<const float4*>(a.norm_ref + (orow - a.out_f32) + 32 * d + 8 * gq);
        const float e0 = v.x - p.x, e1 = v.y - p.y, e2 = v.z - p.z, e3 = v.w - p.w;
        ss += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
      }
    }
  }
  if (a.norm_ref != nullptr) {
    // the other half of the head's channels sits in lane ^ 32; lane half 0 writes the (token, head) partial.  The selection
    // (evt_select_*_sq) adds the H partials of a token in index order: no separate pass over the attention output.
    ss += __shfl_xor(ss, 32, 64);
    if (tr >= 0 && lh == 0) a.norm_parts[((int64_t)(g / a.groups_per_clip) * a.clip_rows + tr) * a.H + h] = ss;
  }
#ifdef EVT_PROF
  WN_TICK(9);   // epilogue
  if (prof_on && lane == 0)
    for (int q = 0; q < 12; ++q) evt_prof_window_buf[q] = prof_acc[q];
#endif
}

template <typename T, bool SPLIT, int NW>
void launch_window_inst(const DenseArgs& a, size_t lds, hipStream_t s) {
  const int waves = (a.N + 31) / 32;
  const dim3 grid((waves + NW - 1) / NW, a.G * a.H);
  EVT_ALLOW_LDS((attn_window_kernel<T, SPLIT, NW>), lds);
  hipLaunchKernelGGL((attn_window_kernel<T, SPLIT, NW>), grid, dim3(64 * NW), lds, s, a);
}

template <typename T>
void launch_window_t(const DenseArgs& a, int split, hipStream_t s) {
  const size_t lds = win_lds(a.N, a.gh + a.gw, (int)sizeof(T), split != 0).total;
  const int waves = (a.N + 31) / 32;
  // Eight waves (two per SIMD) cover up to 256 query rows in one workgroup; groups of up to 128 tokens need four.  (Splitting a
  // pair's query rows over two 4-wave workgroups when there are few pairs -- ViTDet 672^2: 108 on 256 CUs -- was measured
  // slower, 32.0 vs 28.6 us: each workgroup stages the pair's K / V and all rel-pos items itself.)
  const bool four = waves <= 4;
  if (split) {
    if (four) launch_window_inst<T, true, 4>(a, lds, s); else launch_window_inst<T, true, 8>(a, lds, s);
  } else {
    if (four) launch_window_inst<T, false, 4>(a, lds, s); else launch_window_inst<T, false, 8>(a, lds, s);
  }
}

}  // namespace

bool evt_window_fits(int N, int nrel, int store, int split) {
  if (N <= 0 || N > 32 * MAXB) return false;
  const int sb = store == EVT_F32 ? 4 : 2;
  return win_lds(N, nrel, sb, split != 0).total <= (size_t)EVT_LDS_PER_CU;
}

extern "C" int evt_attention_dense_resident(int32_t N, int32_t gh, int32_t gw, int32_t store, int32_t qk_split) {
  return evt_window_fits(N, (gh > 0 && gw > 0) ? gh + gw : 0, store, qk_split) ? 1 : 0;
}

bool evt_launch_window(const DenseArgs& a, int store, int split, hipStream_t s) {
  if (a.product != nullptr || a.a_state != nullptr || a.pv != nullptr || a.out_f32 == nullptr) return false;
  if (!evt_window_fits(a.N, a.gh + a.gw, store, split)) return false;
  switch (store) {
    case EVT_F32: launch_window_t<float>(a, split, s); break;
    case EVT_BF16: launch_window_t<bf16_t>(a, split, s); break;
    case EVT_F16: launch_window_t<f16_t>(a, split, s); break;
    default: return false;
  }
  return true;
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof_window(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_window_buf), sizeof(unsigned long long) * 12);
}
#endif
