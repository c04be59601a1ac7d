// evt_stream_prep.hip -- ONE launch for the three preparations of a gated frame of evt_attention_stream (K9): rel-pos terms of
// every query token (evt_rel_terms, split arithmetic), the key plane (the frame's key rows as bf16 hi / lo fragments) and the value
// delta gate with transposed outputs (evt_v_gate).  All three read the updated token buffer (the value gate also the gate's
// index list) and none reads another's output, so they run as ROLES of one grid: workgroups [0, n_rel) compute rel-pos terms,
// [n_rel, n_rel + n_key) write the key plane, the rest gate the values.  A one-stream ViTDet frame runs four global blocks:
// eight launches and their boundaries less per frame, and the three roles' workgroups fill the chip together instead of one
// after another (9 + 6 + 7 us -> ~11 us at 672^2).  The bodies are the stand-alone kernels' (evt_prep_roles.h): same results.
#include "evt_prep_roles.h"

namespace {

struct PrepArgs {
  const float* qkv; const float* rel_y; const float* rel_x; float* terms;
  void* k_split;
  const int32_t* idx; const int32_t* count; void* v_state; void* v_delta_t; void* v_old_t;
  int B, H, N, D, gh, gw, qw, kcap;
  int n_rel, n_key, rel_x_blocks, key_x_blocks, vg_x_blocks, vg_y_blocks;
  // keys / values: Nk rows of `ksrc` (row stride k_rs floats, keys at k_off, values at v_off): the packed token buffer, or the
  // pooled (B,Nk,2D) buffer of evt_pool_kv
  const float* ksrc; int64_t k_rs; int Nk, k_off, v_off;
};

template <typename T>
__global__ __launch_bounds__(256) void stream_prep_kernel(const PrepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char prep_smem[];
  int w = blockIdx.x;   // role by workgroup range (uniform per workgroup)
  if (w < a.n_rel) {
    evt_rel_terms_mfma_role(a.qkv, a.rel_y, a.rel_x, a.H, a.N, a.D, a.gh, a.gw, a.qw, a.terms, w % a.rel_x_blocks, w / a.rel_x_blocks,
                            reinterpret_cast<float*>(prep_smem));
    return;
  }
  w -= a.n_rel;
  if (w < a.n_key) {
    evt_split_keys_role(a.ksrc, a.k_rs, a.k_off, reinterpret_cast<uint4*>(a.k_split), a.B, a.H, a.Nk, evt_key_blocks(a.Nk, a.gh, a.gw), a.gw, w % a.key_x_blocks, w / a.key_x_blocks,
                        reinterpret_cast<uint4*>(prep_smem));
    return;
  }
  w -= a.n_key;
  const int bx = w % a.vg_x_blocks, r = w / a.vg_x_blocks;
  evt_v_gate_t_role<T>(a.ksrc + a.v_off, a.k_rs, a.idx, a.count, a.Nk, a.D, a.kcap, reinterpret_cast<T*>(a.v_state),
                       reinterpret_cast<T*>(a.v_delta_t), reinterpret_cast<T*>(a.v_old_t), bx, r % a.vg_y_blocks, r / a.vg_y_blocks, prep_smem);
}

}  // namespace

extern "C" int evt_stream_prep(const evt_stream_prep_desc* d, void* stream) {
  EVT_REQUIRE(d != nullptr, EVT_ERR_BAD_ARG, "evt_stream_prep: null descriptor");
  EVT_REQUIRE(d->qkv && d->rel_y && d->rel_x && d->terms && d->k_split && d->idx && d->v_state && d->v_delta_t && d->v_old_t, EVT_ERR_BAD_ARG,
              "evt_stream_prep: null pointer (all three roles are required; use evt_rel_terms / evt_v_gate for a subset)");
  EVT_REQUIRE(d->B >= 0 && d->H > 0 && d->N > 0 && d->gh > 0 && d->gw > 0 && d->qw > 0 && d->N % d->qw == 0 && d->D == d->H * 64, EVT_ERR_BAD_SHAPE,
              "evt_stream_prep: head dim 64 and N = qh * qw required (B=%d H=%d N=%d D=%d qw=%d)", d->B, d->H, d->N, d->D, d->qw);
  EVT_REQUIRE((d->kv == nullptr) == (d->Nk == 0) && d->Nk >= 0 && d->gh * d->gw == (d->kv ? d->Nk : d->N), EVT_ERR_BAD_SHAPE,
              "evt_stream_prep: key grid %dx%d against %d keys (kv and Nk come together: pooled keys)", d->gh, d->gw, d->kv ? d->Nk : d->N);
  EVT_REQUIRE(d->kcap > 0 && (d->kcap % 8) == 0, EVT_ERR_BAD_SHAPE, "evt_stream_prep: kcap=%d must be a positive multiple of 8 (transposed value-gate outputs)", d->kcap);
  const int qh = d->N / d->qw;
  const size_t lds_rel = evt_rel_terms_lds(qh, d->qw, d->gh, d->gw);
  EVT_REQUIRE(lds_rel <= (size_t)EVT_LDS_PER_CU, EVT_ERR_BAD_SHAPE, "evt_stream_prep: grid %dx%d too large", qh, d->qw);
  if (d->B == 0) return EVT_OK;
  const int Nk = d->kv ? d->Nk : d->N;
  const int nkb = evt_key_blocks(Nk, d->gh, d->gw);
  PrepArgs a{d->qkv, d->rel_y, d->rel_x, d->terms, d->k_split, d->idx, d->count, d->v_state, d->v_delta_t, d->v_old_t,
             d->B, d->H, d->N, d->D, d->gh, d->gw, d->qw, d->kcap, 0, 0, qh + d->qw, d->B * nkb, (d->kcap + 63) / 64, d->D / 64,
             d->kv ? d->kv : d->qkv, d->kv ? 2 * (int64_t)d->D : 3 * (int64_t)d->D, Nk, d->kv ? 0 : d->D, d->kv ? d->D : 2 * d->D};
  a.n_rel = a.rel_x_blocks * d->B * d->H;
  a.n_key = a.key_x_blocks * ((d->H + EVT_SKH - 1) / EVT_SKH);
  const int n_vg = a.vg_x_blocks * a.vg_y_blocks * d->B;
  const dim3 grid((unsigned)(a.n_rel + a.n_key + n_vg));
  hipStream_t s = evt_stream(stream);
#define EVT_PREP_LAUNCH(T)                                                                                              \
  do {                                                                                                                   \
    const size_t lds = std::max(std::max(lds_rel, EVT_KEY_PLANE_LDS), evt_vgate_lds<T>());                              \
    EVT_ALLOW_LDS(stream_prep_kernel<T>, lds);                                                                           \
    hipLaunchKernelGGL(stream_prep_kernel<T>, grid, dim3(256), lds, s, a);                                               \
  } while (0)
  switch (d->store) {
    case EVT_F32: EVT_PREP_LAUNCH(float); break;
    case EVT_BF16: EVT_PREP_LAUNCH(bf16_t); break;
    case EVT_F16: EVT_PREP_LAUNCH(f16_t); break;
    default: return evt_fail(EVT_ERR_BAD_DTYPE, "evt_stream_prep: unsupported store dtype %d", (int)d->store);
  }
#undef EVT_PREP_LAUNCH
  return evt_check_launch("evt_stream_prep");
}
