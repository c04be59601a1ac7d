// evt_attn_gated.hip -- K10: the attention of EventfulBlock for at most 256 tokens (every ViViT frame), RESIDENT form: one
// workgroup per (clip, head), first frame or gated frame in ONE launch, value gate included.
//
//   v gate    v~ = round(v[idx]); dv~ = round(v~ - v_ref[idx]); v_old = round(v~ - dv~); v_ref[idx] = v~      modules.py:187-201, 294
//   scores    x[i,j] = (q[i] / scale) . k[j], softmax over j                                                    blocks.py:514-522
//   A gate    a~ = round(softmax(x))[:, idx]; da~ = round(a~ - ref[:, idx]); ref[:, idx] = a~  ("col")          modules.py:187-201
//   A.v       pv += round(a~ . dv~); pv += round(da~ . v_old); out = pv, heads merged                           modules.py:285-295
//   first=1   ref = round(softmax(x)); v_ref = round(v); pv = out = round(ref . v_ref)                          modules.py:183-185, 277-283
//
// Why.  The 32-row kernel (evt_attn_fused.hip) re-reads a head's keys for each of its 7 row tiles and every workgroup lives through
// a chain of dependent round trips (index list -> keys -> scores -> 2-byte gathers of the reference -> ...) that three workgroups
// per CU cannot hide: 0.29 of the HBM roofline, VALU-issue and latency bound (profiles/r05/attn_pmc_counters.txt).  Here a head's
// keys (bf16 hi | lo planes) and its value operands enter LDS ONCE, the workgroup's requests are spread over its life (what the staging
// needs in the prologue, the rest behind the second barrier and in front of pass 1), and behind three workgroup barriers (q fragments
// taken / planes resident / score products done) no wave waits for another:
//
//   * scores TRANSPOSED on v_mfma_f32_32x32x16_bf16 (A = 32 keys from LDS, B = the wave's 32 query rows): a lane owns ONE query
//     row and 16 keys per 32-key block, the 32 x N score block stays in registers, row max / sum are in-lane + one exchange;
//   * the selected columns are NOT gathered: the value operands are scattered into full-length [channel][key] planes with ZERO
//     columns for the keys that are not selected, so the two accumulator products contract over all keys -- the probabilities stay in
//     the accumulator layout of S^T, which IS the B-operand layout of out^T = V^T P^T (as in evt_attn_window.hip); a zero column
//     contributes an exact 0 to the fp32 accumulation, whatever the (finite) probability;
//   * the gate reference is kept TILED: 32 x 32 tiles of (query row tile, key block), inside a tile in the order the lanes hold
//     the values -- (t, lane, n): 16-key half t, lane = 32 (c / 4 % 2) + row, n = 4 (c / 8 % 2) + c % 4 for key-in-block c -- so a
//     wave reads / rewrites a tile with two 1 KB loads / stores (the 32-row kernel: 2-byte gathers and scattered 2-byte stores);
//     matmul_gate.p is exposed as the logical (B,H,N,N) tensor by the host (modules.py of this package);
//   * the value gate (evt_v_gate, a launch of its own before) runs in the staging phase: the head's 64 channels of the selected
//     value rows are read, gated against v_ref and written into the planes directly -- its (B,D,kcap) outputs never exist;
//   * every global access is whole cache lines: q rows, A.v state rows, the next gate's reference and the outputs move between
//     16-byte-per-lane coalesced accesses and the accumulator layout through wave-private LDS blocks inside the K-plane region.
//
// 16-bit store types only (the reference's `matmul_2_cast`: products of bf16 / fp16 operands are exact on the matrix cores, fp32
// accumulate, every rounding point of the reference kept), head dim 64, split-precision scores (q, k as bf16 hi + lo, three MFMAs
// per product: the arithmetic of evt_softmax_av_gated's default mode), no relative position, un-pooled keys.  Everything else stays
// on evt_softmax_av_gated / evt_attention_stream.
#include "evt_common.h"
#include "evt_linear.h"   // split4, bf16x8_t

#ifdef EVT_PROF   // phase timing of wave 0 of one workgroup (scripts/attn_prof.py --gated)
__device__ unsigned long long evt_prof_gated_buf[12];
#define GT_TICK(slot) do { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[slot] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define GT_TICK(slot) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (native vector: arrays of HIP's uint4 struct can end up in scratch memory)

constexpr int DH = 64;
constexpr int KPB = DH + 8;   // bf16 pitch of a K plane row (144 B: conflict-free ds_read_b128 over 16-lane groups)

struct GatedArgs {
  const float* qkv; void* a_tiles; const int32_t* idx; const int32_t* count;
  void* v_state; void* pv; float* out_f32; const float* norm_ref; float* norm_parts;
  int B, H, N, D, kcap, first;
  float scale;
};

// 16-bit store types: two values per 32-bit word, element 0 in the low half
template <typename T> struct H16;
template <> struct H16<bf16_t> {
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {   // round to nearest even (v_cvt_pk_bf16_f32)
    union { bf16x2_t v; uint32_t u; } c;
    c.v = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return c.u;
  }
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ f32x16 mfma(uint4 x, uint4 y, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, x), __builtin_bit_cast(bf16x8_t, y), acc, 0, 0, 0);
  }
};
template <> struct H16<f16_t> {
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    union { f16x2_t v; uint32_t u; } c;
    c.v = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
    return c.u;
  }
  static __device__ __forceinline__ float lo(uint32_t w) { union { uint32_t u; f16x2_t v; } c; c.u = w; return (float)c.v.x; }
  static __device__ __forceinline__ float hi(uint32_t w) { union { uint32_t u; f16x2_t v; } c; c.u = w; return (float)c.v.y; }
  static __device__ __forceinline__ f32x16 mfma(uint4 x, uint4 y, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, x), __builtin_bit_cast(f16x8_t, y), acc, 0, 0, 0);
  }
};

// LDS: K hi | lo planes [NP][KPB] bf16, value planes Vd | Vo [DH][VP] (store type), selection masks [NB][2][2] x 16 bytes.
// VP = NP + 4: VP / 2 words = 2 x odd -- the 8-byte reads of 32 consecutive channel rows cover the 64 banks once.
template <int NB> struct GLds {
  static constexpr int NP = 32 * NB, VP = NP + 4;
  static constexpr size_t k_bytes = (size_t)NP * KPB * 2 * 2;
  static constexpr size_t v_plane = (size_t)DH * VP * 2;
  static constexpr size_t mask_bytes = (size_t)NB * 4 * 16;
  static constexpr size_t total = k_bytes + 2 * v_plane + mask_bytes;
};

// NB: 32-key blocks (= 32-row query tiles) the instantiation holds; a launch with fewer tiles (NT = ceil(N / 32) <= NB) skips the rest.
// FIRST: the first frame of a clip (no old reference, no delta operands: one accumulator product over all keys).
template <typename T, int NB, bool FIRST>
__global__ __launch_bounds__(NB <= 4 ? 256 : 512, NB <= 4 ? 2 : 1) void attn_gated_kernel(const GatedArgs a) {
  typedef H16<T> HT;
  typedef GLds<NB> L;
  constexpr int NW = NB <= 4 ? 4 : 8, NTH = 64 * NW;
  constexpr int NP = L::NP, VP = L::VP;
  constexpr int IT = NP * 16 / NTH;   // 16-byte pieces of the K rows (and of the value rows) per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* const Khi = reinterpret_cast<__bf16*>(smem);
  __bf16* const Klo = Khi + (size_t)NP * KPB;
  uint16_t* const Vd = reinterpret_cast<uint16_t*>(smem + L::k_bytes);
  uint16_t* const Vo = Vd + (size_t)DH * VP;
  uint32_t* const msk = reinterpret_cast<uint32_t*>(smem + L::k_bytes + 2 * L::v_plane);   // [(b, t, lh)][4 words]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int NT = (a.N + 31) >> 5;
  constexpr bool first = FIRST;
  const uint32_t rs_b = (uint32_t)(3 * a.D) * 4u;   // bytes per token row of the packed buffer
  const char* const clip = reinterpret_cast<const char*>(a.qkv + (int64_t)b * a.N * 3 * a.D + h * DH);   // this head's q channels of token 0
  const int cnt = first ? a.N : (a.count ? min(a.count[b], a.kcap) : a.kcap);
#ifdef EVT_PROF
  const bool prof_on = blockIdx.x == gridDim.x / 2 && wave == 0;
  unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif

  // ---- value planes and selection masks start as zeros (the columns of keys that are not selected STAY zero) ----------------
  {
    uint4* z = reinterpret_cast<uint4*>(smem + L::k_bytes);
    constexpr int ZN = (int)((2 * L::v_plane + L::mask_bytes) / 16);
    for (int e = tid; e < ZN; e += NTH) z[e] = make_uint4(0u, 0u, 0u, 0u);
  }

  // ---- requests.  Everything the workgroup reads from HBM / L2 is asked for here, oldest first in the order of its use (vmcnt
  //      retires in order): index list -> q rows -> K rows -> (index list back) value rows + value reference; the gate reference tiles
  //      and the A.v state rows follow behind the second barrier.  Every load instruction of the phase addresses whole cache lines: what a CU pays for a load is the
  //      number of separate lines its 64 lanes touch (q rows read straight into MFMA fragments were 64 lines of 16 useful bytes per
  //      instruction -- a third of all the line requests of the workgroup).
  const int32_t* ix = first ? nullptr : a.idx + (int64_t)b * a.kcap;
  int vj[IT];   // key of this thread's value piece `it` (-1: none)
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int p = (tid + NTH * it) >> 4;
    if (first) vj[it] = p < a.N ? p : -1;
    else { const int j = ix[min(p, a.kcap - 1)]; vj[it] = p < cnt ? j : -1; }   // clamped: branch-free
  }
  const int i0 = wave_s * 32, iq = i0 + lr;
  const bool wave_on = i0 < a.N, q_on = iq < a.N;
  // q rows of the wave, coalesced: 16 lanes x 16 bytes per row, four rows per instruction
  f32x4 qg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (lane >> 4) + 4 * i;
    qg[i] = *reinterpret_cast<const f32x4*>(clip + ((uint32_t)min(i0 + r, a.N - 1) * rs_b + (uint32_t)(lane & 15) * 16u));
  }
  f32x4 kr[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = tid + NTH * it, j = e >> 4, c4 = e & 15;
    const f32x4 x = *reinterpret_cast<const f32x4*>(clip + ((uint32_t)min(j, a.N - 1) * rs_b + (uint32_t)(a.D + c4 * 4) * 4u));
    kr[it] = j < a.N ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // gate reference tiles of this wave's query rows: tile (wave, kb) = 2 KB, piece t of a lane = its 8 keys of MFMA step (kb, t)
  // Padding is neither read nor written: rows past N of the last row tile (their lanes read lane 0's piece, branch-free, and do
  // not store), and the second 16-key piece of the last key block when it holds no key (N % 32 in 1..16) -- at N = 197 that is
  // 18 % of the tile bytes.
  uint4 oa[NB][2];
  char* const tile0 = reinterpret_cast<char*>(a.a_tiles) + ((int64_t)bh * NT * NT + (int64_t)min(wave_s, NT - 1) * NT) * 2048;
  char* const tiles = tile0 + lane * 16;                       // stores (lanes with a query row only)
  const char* const tiles_ld = tile0 + (q_on ? lane : 0) * 16;
  const bool half_empty = (a.N & 31) != 0 && (a.N & 31) <= 16;
  auto piece_on = [&](int kb, int t) __attribute__((always_inline)) { return kb < NT && !(t == 1 && kb == NT - 1 && half_empty); };   // wave-uniform
  f32x4 vr[IT];
  uint2 vs[IT];   // the value reference of the same 4 channels (gated frames)
  T* const vst = reinterpret_cast<T*>(a.v_state) + ((int64_t)b * a.N * a.D + h * DH);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c4 = (tid + NTH * it) & 15, jc = max(vj[it], 0);
    vr[it] = *reinterpret_cast<const f32x4*>(clip + ((uint32_t)jc * rs_b + (uint32_t)(2 * a.D + c4 * 4) * 4u));
    vs[it] = make_uint2(0u, 0u);
    if (!first) vs[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(vst) + (uint32_t)(jc * a.D + c4 * 4) * 2u);
  }
  GT_TICK(0);   // zero fill + requests issued

  // ---- q rows -> MFMA fragments through a wave-private LDS block (inside the K plane region, which is written behind the barrier
  //      below): rows at pitch 68 floats, the fragment reads of 16 rows cover the 64 banks once.  q / self.scale (blocks.py:514; a
  //      power-of-two scale makes the reciprocal multiply exact), split into bf16 hi | lo. ---------------------------------------
  const float inv_scale = 1.0f / a.scale;
  const bool pow2 = (inv_scale * a.scale == 1.0f) && ((__float_as_uint(a.scale) & 0x007fffffu) == 0u);
  bf16x8_t qh[4], ql[4];
  if (wave_on) {   // (a wave without query rows has no staging block: the region only holds NB of them)
    constexpr int QP = DH + 4;
    static_assert((size_t)NB * 32 * QP * 4 <= L::k_bytes, "the q staging blocks fit the K plane region");
    float* qst = reinterpret_cast<float*>(smem) + (size_t)wave * 32 * QP;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(qst + ((lane >> 4) + 4 * i) * QP + 4 * (lane & 15)) = qg[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    f32x4 qraw[8];   // channels 16 s + 8 lh + {0..3, 4..7} (s = piece / 2)
#pragma unroll
    for (int m = 0; m < 8; ++m) qraw[m] = *reinterpret_cast<const f32x4*>(qst + lr * QP + 16 * (m >> 1) + 8 * lh + 4 * (m & 1));
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 q0 = make_float4(qraw[2 * s4][0], qraw[2 * s4][1], qraw[2 * s4][2], qraw[2 * s4][3]);
      float4 q1 = make_float4(qraw[2 * s4 + 1][0], qraw[2 * s4 + 1][1], qraw[2 * s4 + 1][2], qraw[2 * s4 + 1][3]);
      if (pow2) {
        q0.x *= inv_scale; q0.y *= inv_scale; q0.z *= inv_scale; q0.w *= inv_scale;
        q1.x *= inv_scale; q1.y *= inv_scale; q1.z *= inv_scale; q1.w *= inv_scale;
      } else {
        q0.x /= a.scale; q0.y /= a.scale; q0.z /= a.scale; q0.w /= a.scale;
        q1.x /= a.scale; q1.y /= a.scale; q1.z /= a.scale; q1.w /= a.scale;
      }
      bf16x4_t h0, l0, h1, l1;
      split4(q0, &h0, &l0);
      split4(q1, &h1, &l1);
      qh[s4] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      ql[s4] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
  __syncthreads();   // zero fill complete; every wave holds its q fragments (the K planes overwrite the staging blocks)
  GT_TICK(1);   // q fragments + barrier

  // ---- K rows -> bf16 hi | lo planes (key-major) -----------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = tid + NTH * it, j = e >> 4, c4 = e & 15;
    bf16x4_t hi, lo;
    split4(make_float4(kr[it][0], kr[it][1], kr[it][2], kr[it][3]), &hi, &lo);
    *reinterpret_cast<bf16x4_t*>(Khi + (size_t)j * KPB + c4 * 4) = hi;
    *reinterpret_cast<bf16x4_t*>(Klo + (size_t)j * KPB + c4 * 4) = lo;
  }
  // ---- value gate of the selected keys (evt_v_gate's arithmetic), results scattered into column `key` of the planes ------------
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c4 = (tid + NTH * it) & 15, j = vj[it];
    if (j >= 0) {
      const float x0 = vr[it][0], x1 = vr[it][1], x2 = vr[it][2], x3 = vr[it][3];
      const uint32_t n01 = HT::pack2(x0, x1), n23 = HT::pack2(x2, x3);   // v~ = round(v)
      uint32_t d01 = n01, d23 = n23, o01 = 0u, o23 = 0u;
      if (!first) {
        const float v0 = HT::lo(n01), v1 = HT::hi(n01), v2 = HT::lo(n23), v3 = HT::hi(n23);
        d01 = HT::pack2(v0 - HT::lo(vs[it].x), v1 - HT::hi(vs[it].x));   // dv~ = round(v~ - v_ref)
        d23 = HT::pack2(v2 - HT::lo(vs[it].y), v3 - HT::hi(vs[it].y));
        o01 = HT::pack2(v0 - HT::lo(d01), v1 - HT::hi(d01));             // v_old = round(v~ - dv~), modules.py:294
        o23 = HT::pack2(v2 - HT::lo(d23), v3 - HT::hi(d23));
      }
      *reinterpret_cast<uint2*>(reinterpret_cast<char*>(vst) + (uint32_t)(j * a.D + c4 * 4) * 2u) = make_uint2(n01, n23);   // v_ref[key] = v~
      uint16_t* pd = Vd + (size_t)(c4 * 4) * VP + j;
      pd[0] = (uint16_t)d01; pd[VP] = (uint16_t)(d01 >> 16); pd[2 * VP] = (uint16_t)d23; pd[3 * VP] = (uint16_t)(d23 >> 16);
      if (!first) {
        uint16_t* po = Vo + (size_t)(c4 * 4) * VP + j;
        po[0] = (uint16_t)o01; po[VP] = (uint16_t)(o01 >> 16); po[2 * VP] = (uint16_t)o23; po[3 * VP] = (uint16_t)(o23 >> 16);
        if (c4 == 0) {   // selection mask of the key, in the packing of the lanes' probability words (see the tile layout)
          const int kb = j >> 5, c = j & 31, t = c >> 4, mh = (c >> 2) & 1, n = (c & 3) + ((c >> 3) & 1) * 4;
          atomicOr(msk + ((kb * 2 + t) * 2 + mh) * 4 + (n >> 1), 0xffffu << (16 * (n & 1)));
        }
      }
    }
  }
  __syncthreads();   // the last workgroup barrier: K planes, value planes and masks are resident
  GT_TICK(2);   // planes written + barrier
  // Second wave of requests, for the data that is not needed before pass 2 / the epilogue: the gate reference tiles of this wave's
  // query rows and its A.v state rows.  Asked for here instead of in the prologue they fly during the score products, the softmax and
  // pass 1 -- a CU sustains only so many outstanding lines (~45 KB in flight: the prologue's 255 KB took 11 us at 22 GB/s while the
  // memory pipe idled through the 12 us of arithmetic that followed), so spreading the requests over the workgroup's life is what
  // raises the bytes per second, not issuing them earlier.
  // A.v state rows of the epilogue, coalesced (8 lanes x 16 bytes per row of this head, 8 rows per instruction): they reach the
  // lanes that own the channels through the wave's LDS block (below).  Read straight into the accumulator layout -- 4 channels of
  // 32 different rows per instruction -- every 8-byte piece was a line request of its own: with the next gate's reference and the
  // state stores, 63 % of all the line requests of a workgroup.
  T* const pvb = reinterpret_cast<T*>(a.pv) + ((int64_t)b * a.N * a.D + h * DH);
  u32x4 pvc[4];
  if (!first) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pvc[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(pvb) + (uint32_t)(min(i0 + (lane >> 3) + 8 * i, a.N - 1) * a.D + (lane & 7) * 8) * 2u);
  }
  // (the tiles go out behind the state rows: vmcnt retires in order, and the state rows are wanted first -- behind the next barrier)
  if (!first) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (piece_on(kb, t)) oa[kb][t] = *reinterpret_cast<const uint4*>(tiles_ld + kb * 2048 + t * 1024);
  }

  // ---- S^T = K (q / scale)^T: block kb = keys 32 kb .. + 31; lane (lr, lh), register r = 4 g + e holds key 32 kb + 8 g + 4 lh + e
  //      of query row lr --------------------------------------------------------------------------------------------------------
  f32x16 S[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) S[kb][r] = 0.f;
    if (kb < NT && wave_on) {
      const __bf16* kh_row = Khi + (size_t)(32 * kb + lr) * KPB + 8 * lh;
      const __bf16* kl_row = Klo + (size_t)(32 * kb + lr) * KPB + 8 * lh;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(kh_row + 16 * s);
        const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(kl_row + 16 * s);
        S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], S[kb], 0, 0, 0);
        S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], S[kb], 0, 0, 0);
        S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], S[kb], 0, 0, 0);
      }
    }
  }
  // Every wave is done with the K planes: from here on the region is ONE PRIVATE BLOCK PER WAVE (32 rows x 288 bytes), through which
  // the epilogue's rows travel between whole-line global accesses and the accumulator layout.  No wave waits for another again.
  __syncthreads();
  if (!wave_on) return;
  constexpr int EPB = 144, EPF = 272;   // row pitch of the block in bytes: 64 store-type elements + 8, 64 floats + 4
  static_assert(32 * EPF <= 32 * KPB * 4 && (size_t)NB * 32 * KPB * 4 <= L::k_bytes, "one epilogue block per wave inside the K region");
  char* const eb = reinterpret_cast<char*>(smem) + (size_t)wave * 32 * KPB * 4;
  if (!first) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(eb + ((lane >> 3) + 8 * i) * EPB + (lane & 7) * 16) = pvc[i];
  }
  GT_TICK(3);   // S^T products + barrier

  // ---- softmax of the lane's row (its other half sits in lane ^ 32) --------------------------------------------------------
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
    if (kb < NT) {
      if (32 * kb + 32 > a.N) {   // only the last block has keys past N (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * kb + 8 * (r >> 2) + 4 * lh + (r & 3);
          S[kb][r] = key < a.N ? S[kb][r] : -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[kb][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
    if (kb < NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f((S[kb][r] - mx) * 1.44269504088896340736f);   // exp2(-inf) = 0 for masked keys
        S[kb][r] = p;
        sum += p;
      }
    }
  sum += __shfl_xor(sum, 32, 64);
  const float rinv = 1.0f / sum;   // one division per row: e * (1 / sum) is within 1 ulp of e / sum before the rounding to the store type
  GT_TICK(4);   // softmax

  // round(softmax) of the lane's 8 keys of MFMA step (kb, t), packed two per word: from here on the probabilities take half the
  // registers the scores did (the row sum above is that of the unrounded exponentials, as in the reference)
  uint4 AN[NB][2];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (kb < NT)
        AN[kb][t] = make_uint4(HT::pack2(S[kb][8 * t + 0] * rinv, S[kb][8 * t + 1] * rinv), HT::pack2(S[kb][8 * t + 2] * rinv, S[kb][8 * t + 3] * rinv),
                               HT::pack2(S[kb][8 * t + 4] * rinv, S[kb][8 * t + 5] * rinv), HT::pack2(S[kb][8 * t + 6] * rinv, S[kb][8 * t + 7] * rinv));

  // the next gate's reference rows of this head, coalesced (16 lanes x 16 bytes per row, 4 rows per instruction)
  f32x4 nrc[8];
  auto load_nref = [&]() __attribute__((always_inline)) {
    const char* nb = reinterpret_cast<const char*>(a.norm_ref + ((int64_t)b * a.N * a.D + h * DH));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      nrc[i] = *reinterpret_cast<const f32x4*>(nb + (uint32_t)(min(i0 + (lane >> 4) + 4 * i, a.N - 1) * a.D + (lane & 15) * 4) * 4u);
  };

  if (a.norm_ref != nullptr) load_nref();   // (the score registers are free again: lands during the two passes)

  // ---- pass 1: O1^T = Vd^T a~^T.  MFMA step (kb, t) contracts keys 32 kb + 16 t + 4 lh' + {0..3, 8..11} -- the keys of a lane's
  //      registers 8 t .. 8 t + 7 of block kb -- against the plane's row pieces at the same keys.  First frame: the probabilities
  //      are also the new gate reference (their packing IS the tile layout). --------------------------------------------------
  auto vfrag = [&](const uint16_t* plane, int d, int k0) __attribute__((always_inline)) {
    const uint16_t* p = plane + (size_t)(32 * d + lr) * VP + k0;
    const uint2 x = *reinterpret_cast<const uint2*>(p), y = *reinterpret_cast<const uint2*>(p + 8);
    return make_uint4(x.x, x.y, y.x, y.y);
  };
  f32x16 O[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
    if (kb < NT) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (piece_on(kb, t)) {
          const int k0 = 32 * kb + 16 * t + 4 * lh;
#pragma unroll
          for (int d = 0; d < 2; ++d) O[d] = HT::mfma(vfrag(Vd, d, k0), AN[kb][t], O[d]);
          if (first && q_on) *reinterpret_cast<uint4*>(tiles + kb * 2048 + t * 1024) = AN[kb][t];
        }
    }
  GT_TICK(5);   // pass 1
  // round(a~ . dv~), packed: lane (query lr, lh), word 2 g + w of tile d = channels 32 d + 8 g + 4 lh + 2 w, + 1
  uint32_t r1[2][8];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int w = 0; w < 8; ++w) r1[d][w] = HT::pack2(O[d][2 * w], O[d][2 * w + 1]);

  // ---- pass 2 (gated frames): da~ = round(a~ - ref), O2^T = Vo^T da~^T, ref = selected ? a~ : ref ------------------------------
  if (!first) {
    // The epilogue's loads (the next gate's reference: requested in front of pass 1) are claimed HERE, while no store is in
    // flight: the tile stores below sit in exec-masked blocks hipcc cannot count, so the first wait behind them is vmcnt(0) -- in the
    // epilogue that would be a whole store round trip in the open.
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (a.norm_ref != nullptr) asm volatile("" : "+v"(nrc[4 * d + g]));
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
      if (kb < NT) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (piece_on(kb, t)) {
          const uint4 an = AN[kb][t], old = oa[kb][t];
          const uint4 m = *reinterpret_cast<const uint4*>(msk + ((kb * 2 + t) * 2 + lh) * 4);
          uint4 ad, nw;
          ad.x = HT::pack2(HT::lo(an.x) - HT::lo(old.x), HT::hi(an.x) - HT::hi(old.x));
          ad.y = HT::pack2(HT::lo(an.y) - HT::lo(old.y), HT::hi(an.y) - HT::hi(old.y));
          ad.z = HT::pack2(HT::lo(an.z) - HT::lo(old.z), HT::hi(an.z) - HT::hi(old.z));
          ad.w = HT::pack2(HT::lo(an.w) - HT::lo(old.w), HT::hi(an.w) - HT::hi(old.w));
          nw.x = (an.x & m.x) | (old.x & ~m.x);
          nw.y = (an.y & m.y) | (old.y & ~m.y);
          nw.z = (an.z & m.z) | (old.z & ~m.z);
          nw.w = (an.w & m.w) | (old.w & ~m.w);
          const int k0 = 32 * kb + 16 * t + 4 * lh;
#pragma unroll
          for (int d = 0; d < 2; ++d) O[d] = HT::mfma(vfrag(Vo, d, k0), ad, O[d]);
          if (q_on) *reinterpret_cast<uint4*>(tiles + kb * 2048 + t * 1024) = nw;
        }
      }
  }
  GT_TICK(6);   // pass 2

  // ---- epilogue: pv = round(round(pv + round(O1)) + round(O2)) (first frame: pv = round(O1)); out = pv; heads merged.  Lane (query
  //      lr, lh) owns channels 32 d + 8 g + 4 lh + 0..3; rows move between that layout and whole-line global accesses through `eb`.
  auto wave_sync = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); };
  const uint32_t ech = (uint32_t)(4 * lh);
  uint2 pvr[2][4];
  f32x4 nrr[2][4];
  if (!first) {
    wave_sync();
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) pvr[d][g] = *reinterpret_cast<const uint2*>(eb + lr * EPB + (32 * d + 8 * g + ech) * 2);
  }
  if (a.norm_ref != nullptr) {
    wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(eb + ((lane >> 4) + 4 * i) * EPF + (lane & 15) * 16) = nrc[i];
    wave_sync();
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) nrr[d][g] = *reinterpret_cast<const f32x4*>(eb + lr * EPF + (32 * d + 8 * g + ech) * 4);
  }
  wave_sync();
  float ss = 0.f;
  uint2 res[2][4];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t w0 = r1[d][2 * g], w1 = r1[d][2 * g + 1];
      if (!first) {
        w0 = HT::pack2(HT::lo(pvr[d][g].x) + HT::lo(w0), HT::hi(pvr[d][g].x) + HT::hi(w0));
        w1 = HT::pack2(HT::lo(pvr[d][g].y) + HT::lo(w1), HT::hi(pvr[d][g].y) + HT::hi(w1));
        const uint32_t s0 = HT::pack2(O[d][4 * g + 0], O[d][4 * g + 1]), s1 = HT::pack2(O[d][4 * g + 2], O[d][4 * g + 3]);
        w0 = HT::pack2(HT::lo(w0) + HT::lo(s0), HT::hi(w0) + HT::hi(s0));
        w1 = HT::pack2(HT::lo(w1) + HT::lo(s1), HT::hi(w1) + HT::hi(s1));
      }
      res[d][g] = make_uint2(w0, w1);
      *reinterpret_cast<uint2*>(eb + lr * EPB + (32 * d + 8 * g + ech) * 2) = res[d][g];
      if (a.norm_ref != nullptr) {
        const float e0 = HT::lo(w0) - nrr[d][g][0], e1 = HT::hi(w0) - nrr[d][g][1], e2 = HT::lo(w1) - nrr[d][g][2], e3 = HT::hi(w1) - nrr[d][g][3];
        ss += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
      }
    }
  wave_sync();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 3) + 8 * i;
    if (i0 + row < a.N)
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(pvb) + (uint32_t)((i0 + row) * a.D + (lane & 7) * 8) * 2u) =
          *reinterpret_cast<const uint4*>(eb + row * EPB + (lane & 7) * 16);
  }
  if (a.out_f32 != nullptr) {   // launch-uniform; NULL: the caller reads the (identical) values from the A.v state
    wave_sync();
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(eb + lr * EPF + (32 * d + 8 * g + ech) * 4) =
            (f32x4){HT::lo(res[d][g].x), HT::hi(res[d][g].x), HT::lo(res[d][g].y), HT::hi(res[d][g].y)};
    wave_sync();
    char* const ob = reinterpret_cast<char*>(a.out_f32 + ((int64_t)b * a.N * a.D + h * DH));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (lane >> 4) + 4 * i;
      if (i0 + row < a.N)
        *reinterpret_cast<f32x4*>(ob + (uint32_t)((i0 + row) * a.D + (lane & 15) * 4) * 4u) = *reinterpret_cast<const f32x4*>(eb + row * EPF + (lane & 15) * 16);
    }
  }
  if (a.norm_ref != nullptr) {
    // the other half of the head's channels sits in lane ^ 32; lane half 0 writes the (token, head) partial (evt_select_*_sq adds
    // the H partials of a token in index order)
    ss += __shfl_xor(ss, 32, 64);
    if (q_on && lh == 0) a.norm_parts[((int64_t)b * a.N + iq) * a.H + h] = ss;
  }
#ifdef EVT_PROF
  GT_TICK(7);   // epilogue
  if (prof_on && lane == 0)
    for (int q = 0; q < 12; ++q) evt_prof_gated_buf[q] = prof_acc[q];
#endif
}

template <typename T, int NB>
void launch_gated_inst(const GatedArgs& a, hipStream_t s) {
  constexpr size_t lds = GLds<NB>::total;
  if (a.first) {
    EVT_ALLOW_LDS((attn_gated_kernel<T, NB, true>), lds);
    hipLaunchKernelGGL((attn_gated_kernel<T, NB, true>), dim3(a.B * a.H), dim3(NB <= 4 ? 256 : 512), lds, s, a);
  } else {
    EVT_ALLOW_LDS((attn_gated_kernel<T, NB, false>), lds);
    hipLaunchKernelGGL((attn_gated_kernel<T, NB, false>), dim3(a.B * a.H), dim3(NB <= 4 ? 256 : 512), lds, s, a);
  }
}

template <typename T>
void launch_gated(const GatedArgs& a, hipStream_t s) {
  const int nt = (a.N + 31) >> 5;
  if (nt <= 2) launch_gated_inst<T, 2>(a, s);
  else if (nt <= 4) launch_gated_inst<T, 4>(a, s);
  else if (nt <= 7) launch_gated_inst<T, 7>(a, s);
  else launch_gated_inst<T, 8>(a, s);
}

}  // namespace

extern "C" int evt_attention_gated_fits(int32_t N, int32_t D, int32_t H, int32_t store, int32_t qk_split) {
  return (N > 0 && N <= 256 && H > 0 && D == 64 * H && (store == EVT_BF16 || store == EVT_F16) && qk_split != 0) ? 1 : 0;
}

extern "C" int64_t evt_attention_gated_tile_bytes(int32_t B, int32_t H, int32_t N) {
  if (B < 0 || H <= 0 || N <= 0) return -1;
  const int64_t nt = (N + 31) >> 5;
  return (int64_t)B * H * nt * nt * 2048;
}

extern "C" int evt_attention_gated(const evt_attn_gated_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_attention_gated: null descriptor");
  EVT_REQUIRE(d->qkv && d->a_tiles && d->v_state && d->pv, EVT_ERR_BAD_ARG, "evt_attention_gated: null pointer");
  EVT_REQUIRE(d->B >= 0 && d->H > 0 && d->N > 0 && d->scale > 0.f, EVT_ERR_BAD_ARG, "evt_attention_gated: bad sizes");
  EVT_REQUIRE(evt_attention_gated_fits(d->N, d->D, d->H, d->store, d->qk_split), EVT_ERR_BAD_SHAPE,
              "evt_attention_gated: needs head dim 64, N <= 256, a 16-bit store type and split-precision scores (N=%d D=%d H=%d store=%d "
              "qk_split=%d); use evt_softmax_av_gated / evt_attention_stream", d->N, d->D, d->H, d->store, d->qk_split);
  if (!d->first)
    EVT_REQUIRE(d->idx != nullptr && d->kcap > 0 && d->kcap <= d->N, EVT_ERR_BAD_ARG, "evt_attention_gated: a gated frame needs idx and 0 < kcap <= N (kcap=%d)", d->kcap);
  EVT_REQUIRE((d->norm_ref == nullptr) == (d->norm_parts == nullptr), EVT_ERR_BAD_ARG, "evt_attention_gated: norm_ref / norm_parts come together");
  // 32-bit byte offsets inside a clip's slice of the packed buffer and of the (N, D) states
  EVT_REQUIRE((int64_t)d->N * 3 * d->D * 4 < (1ll << 32), EVT_ERR_BAD_SHAPE, "evt_attention_gated: a clip's token buffer exceeds 4 GB");
  if (d->B == 0) return EVT_OK;
  GatedArgs a{d->qkv, d->a_tiles, d->idx, d->count, d->v_state, d->pv, d->out_f32, d->norm_ref, d->norm_parts,
              d->B, d->H, d->N, d->D, d->kcap, d->first ? 1 : 0, d->scale};
  hipStream_t s = evt_stream(stream);
  if (d->store == EVT_BF16) launch_gated<bf16_t>(a, s); else launch_gated<f16_t>(a, s);
  return evt_check_launch("evt_attention_gated");
}

#ifdef EVT_PROF
extern "C" __attribute__((visibility("default"))) int evt_debug_prof_gated(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evt_prof_gated_buf), sizeof(unsigned long long) * 12);
}
#endif
