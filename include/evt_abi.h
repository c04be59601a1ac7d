/*
 * evt_abi.h -- C ABI of libevt_hip.so: hand-written gfx950 (CDNA4) HIP kernels for the Eventful
 * Transformer gated-token inference path.
 *
 * The reference (WISION-Lab/eventful-transformer) has NO FFI boundary: its "plugin API" for this
 * path is the Python class API of package `eventful_transformer` and everything below that is a
 * stock ATen op.  This header is therefore the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md); each entry point cites the reference code (file:line under /root/reference)
 * whose ATen op sequence it replaces.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm tensors).  The library
 *     never allocates, frees or retains device memory and keeps no global state.
 *   - `stream` is a hipStream_t passed as void*.  Calls only enqueue work; they never synchronise.
 *   - Return value: EVT_OK (0) or a negative evt_status.  Nothing throws or exits.  A failing call
 *     leaves a message readable through evt_last_error_string() (thread-local).
 *   - Tokens are row-major fp32: (B, N, D) with a token = one contiguous row.
 *   - Index lists are int32, ASCENDING, laid out (B, kcap); `count` (nullable) holds the number of
 *     valid entries per clip, on the device, for data-dependent selections (threshold policy).
 *     count == NULL means every clip has exactly kcap entries (top-k).
 *   - "store type" (evt_dtype) is the type the reference holds a tensor in after
 *     `_cast_matmul_2` (blocks.py:183-189): fp32, or bf16/fp16 with round-to-nearest-even at
 *     exactly the points where the reference rounds.
 */
#ifndef EVT_ABI_H
#define EVT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVT_ABI_VERSION 9   /* 2: split-K workspace fields, evt_qk_desc.split, evt_attention_dense, workspace query;
                                 3: evt_split_weights(rows, cols) writes the hl32 layout, evt_split_weights_bytes;
                                 4: evt_rel_terms, evt_softmax_av_desc.rel_terms, evt_linear_desc.a_bf16,
                                    evt_gated_linear_big_tile;
                                 5: evt_attention_stream (+ its k_split workspace), evt_attn_dense_desc.qk_split, the `split`
                                    argument of evt_rel_terms, embedded selection (sel_* fields of evt_linear_desc /
                                    evt_mlp_desc, evt_gated_linear_embeds_select);
                                 6: evt_attention_stream_lds_bytes (shape-only query; evt_attention_stream now answers
                                    EVT_ERR_BAD_SHAPE instead of a launch error when its tile does not fit a CU's LDS);
                                 7: evt_row_pass_ord (the policies' `order` argument: L1 / L2 / L-infinity delta norms);
                                    evt_stream_prep + evt_attn_stream_desc.k_split_ready (rel-pos terms, key plane and value gate
                                    of a gated frame in one launch); evt_attn_dense_desc.norm_ref / norm_parts +
                                    evt_attention_dense_resident (the projection gate's delta norm from the attention epilogue of
                                    the windowed blocks);
                                 8: evt_attn_stream_desc.kv / Nk, evt_stream_prep_desc.kv / Nk (pooled keys and values in
                                    evt_attention_stream and its preparation launch); REMOVED: the embedded selection of ABI 5
                                    (sel_* fields, evt_gated_linear_embeds_select) -- measured slower than the selection launch
                                    in every configuration;
                                 9: evt_attention_gated (+ _fits, _tile_bytes): the attention of EventfulBlock for <= 256 tokens,
                                    one workgroup per (clip, head), value gate included, gate reference in a tiled layout;
                                    evt_gate_cols / evt_scatter_cols / evt_gate_rows_any / evt_move_rows_any (stand-alone gates and
                                    buffers of any structure and element type), evt_gather_rows_map / evt_scatter_rows_map (window
                                    partition with padding), evt_ats_scores / evt_ats_stabilize (adaptive token sampling) */

/* Exported symbols (the library is built with -fvisibility=hidden). */
#if defined(__GNUC__)
#define EVT_API __attribute__((visibility("default")))
#else
#define EVT_API
#endif

typedef enum evt_status {
  EVT_OK = 0,
  EVT_ERR_BAD_ARG = -1,     /* null pointer, non-positive size, unsupported combination       */
  EVT_ERR_BAD_SHAPE = -2,   /* shape outside what the kernels support (see each entry point)  */
  EVT_ERR_BAD_DTYPE = -3,
  EVT_ERR_HIP = -4          /* hipGetLastError() after a launch; text in evt_last_error_string */
} evt_status;

typedef enum evt_dtype { EVT_F32 = 0, EVT_BF16 = 1, EVT_F16 = 2 } evt_dtype;

typedef enum evt_act { EVT_ACT_NONE = 0, EVT_ACT_GELU_ERF = 1 } evt_act;

EVT_API int evt_version(void);
EVT_API const char* evt_last_error_string(void);
/* Compile-time target of the embedded code object, e.g. "gfx950". */
EVT_API const char* evt_target_arch(void);
/* ABI 9.  Compute units the calling thread's persistent launches (the 256-row gated linear, evt_attention_stream's tile choice) size their
 * grids for: set it to the number of CUs in a stream's CU mask (hipExtStreamCreateWithCUMask) before launching into that stream, 0 = the
 * whole device (default).  Thread-local; nothing else changes -- the library never creates streams. */
EVT_API int evt_set_cu_budget(int32_t cus);

/* ABI 6.  Reads `bytes` (16-byte aligned pointer) of a read-only operand -- a CountedLinear's weight planes (evt_split_weights) --
 * into the memory-side cache ahead of the launch that streams them; `sink` = 4 writable bytes (never written in practice: it keeps
 * the loads).  No reference counterpart: the reference's per-layer weights (blocks.py:102-116) are re-read from HBM by every frame of
 * one video stream, and a 6-14 us launch pays for cold reads.  Issue it on a side stream; results are unaffected. */
EVT_API int evt_prefetch(const void* ptr, int64_t bytes, void* sink, void* stream);

/* ABI 6.  One-shot rider: the NEXT evt_select_* launch issued from this host thread (any of the four) also reads [ptr, ptr + bytes)
 * like evt_prefetch, with extra workgroups of its own grid -- the selection of one video stream is a single workgroup for ~5 us, so
 * the riders cost no launch and no time.  The caller arms it with the weight planes of a gated linear a few launches ahead
 * (blocks.py:430-450: MLP-1 behind the projection gate, MLP-2 and the next block's QKV behind the MLP gate).  A second call before
 * the launch arms a second range.  Results are unaffected. */
EVT_API int evt_select_prefetch_next(const void* ptr, int64_t bytes, void* sink);

/* ------------------------------------------------------------------------------------------ *
 * K0/K1a  Row pass: [residual add] -> [LayerNorm] -> [delta-norm against a gate reference].
 *
 *   s = x (+ res[row % res_rows])         written to sum_out   if non-null  (res_rows == 0: res[row];
 *                                         res_rows == N broadcasts a (1,N,D) position encoding)
 *   c = ln_w ? LN(s; ln_w, ln_b, eps) : s written to c_out     if non-null
 *   norms[row] = || c - p[row] ||_2       written to norms     if non-null  (p == NULL: || c ||_2)
 *
 * Replaces: nn.LayerNorm (blocks.py:460,444; eps 1e-6 at blocks.py:23), CountedAdd residual
 * (blocks.py:436,448; position encoding utils.py:66), and `c - self.p` + `vector_norm` of the gate/policy
 * (modules.py:149, policies.py:63 / :28).  rows = B*N, D % 4 == 0, D <= 4096.
 * ------------------------------------------------------------------------------------------ */
EVT_API int evt_row_pass(const float* x, const float* res, int res_rows, float* sum_out,
                         const float* ln_w, const float* ln_b, float eps, float* c_out,
                         const float* p, float* norms, int rows, int D, void* stream);
/* ABI 7.  The same pass with the policy's norm order (`TokenNorm*(order=...)`, policies.py:11,28,44,63,76 ->
 * torch.linalg.vector_norm(x, ord=order)): norms[row] = || c - p[row] ||_order for order 2, 1 or infinity. */
enum evt_norm_order { EVT_NORM_LINF = 0, EVT_NORM_L1 = 1, EVT_NORM_L2 = 2 };
EVT_API int evt_row_pass_ord(const float* x, const float* res, int res_rows, float* sum_out,
                             const float* ln_w, const float* ln_b, float eps, float* c_out,
                             const float* p, float* norms, int rows, int D, int order, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K1  Token selection from per-token delta norms.  One workgroup per clip; norms staged in LDS;
 *     radix select of the k-th largest key, then a single-pass ordered compaction (per-thread chunk counts + workgroup scan) in index order.
 *
 * evt_select_topk: idx[b, 0..k) = the k tokens with the largest norms, ASCENDING; ties at the
 *   k-th value go to the LOWEST index.  Replaces `vector_norm(...).topk(k, sorted=False)[1]`
 *   (policies.py:63, :93-95), whose order and tie-break are implementation-defined.
 * evt_select_threshold: idx[b, 0..count[b]) = tokens with norm > threshold, ascending;
 *   count[b] written on the device (no host sync).  Replaces `.gt(thr).nonzero()`
 *   (policies.py:28-32, a host sync in the reference).  kcap (row stride of idx) must be >= N.
 *   rest (nullable, (B, N)): the COMPLEMENT list (unselected tokens, ascending, N - count entries per clip),
 *   a by-product of the same compaction; evt_qk uses it to skip rows it has already rewritten.
 * N <= 16384.
 * ------------------------------------------------------------------------------------------ */
EVT_API int evt_select_topk(const float* norms, int B, int N, int k, int32_t* idx, int32_t* rest, void* stream);
EVT_API int evt_select_threshold(const float* norms, int B, int N, float threshold, int kcap,
                         int32_t* idx, int32_t* count, int32_t* rest, void* stream);
/* The same selections from PARTIAL SUMS OF SQUARES: norm[b,i] = sqrt(sum_p sq_parts[(b*N + i)*parts + p]), the parts added
 * in index order.  evt_softmax_av_gated can emit ||attention output - projection_gate.p||^2 per head (norm_parts), which
 * makes the projection gate's delta-norm pass over the attention output (modules.py:149 + policies.py:63) unnecessary. */
EVT_API int evt_select_topk_sq(const float* sq_parts, int parts, int B, int N, int k, int32_t* idx, int32_t* rest, void* stream);
EVT_API int evt_select_threshold_sq(const float* sq_parts, int parts, int B, int N, float threshold, int kcap,
                                    int32_t* idx, int32_t* count, int32_t* rest, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K2  Gate gather + reference update for the selected tokens (rows):
 *       c_tilde[b,i,:] = c[b, idx[b,i], :]            (if c_tilde non-null)
 *       e_tilde[b,i,:] = c[b, idx[b,i], :] - p[...]   (if e_tilde non-null; TokenDeltaGate)
 *       p[b, idx[b,i], :] = c[b, idx[b,i], :]         (if update_p)
 * Replaces gather + scatter_ of TokenGate/TokenDeltaGate.forward_incremental
 * (modules.py:150-152, 196-200).  c, p: (B,N,D); c_tilde/e_tilde: (B,kcap,D).  D % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
EVT_API int evt_gate_gather_update(const float* c, float* p, const int32_t* idx, const int32_t* count,
                           int B, int N, int D, int kcap, float* c_tilde, float* e_tilde,
                           int update_p, void* stream);

/* TokenBuffer incremental update for callers that already hold compact rows:
 *   b[b, idx[b,i], :] = x[b,i,:]   (modules.py:86-97, structure="row").  F % 4 == 0. */
EVT_API int evt_scatter_rows(const float* x, float* buf, const int32_t* idx, const int32_t* count,
                     int B, int N, int F, int kcap, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K3  Gated linear:  out[orow(m), :] = act( A[arow(m), :] . W^T + bias )  for m = (b, i),
 *     i < count[b], with fp32 MFMA (v_mfma_f32_32x32x2_f32; exact fp32 products/accumulate).
 *       arow(b,i) = b*a_rows + (a_idx ? a_idx[b*kcap+i] : i)
 *       orow(b,i) = b*o_rows + (o_idx ? o_idx[b*kcap+i] : i)
 *     A rows are gathered through a_idx while the A tile is staged into LDS (no compact copy);
 *     output rows are scattered through o_idx in the epilogue (TokenBuffer update fused).
 *     If p_upd is non-null the kernel also performs the gate reference update
 *     p_upd[arow(m), :] = A[arow(m), :] (K2 fused; requires a_idx).
 * Replaces CountedLinear.forward on the gathered rows (counting.py:156-162 called from
 * blocks.py:462,433,244-245), the TokenBuffer scatter_ (modules.py:96) and nn.GELU (blocks.py:114).
 * W: (Nout, K) row-major as in the reference state_dict; K % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_linear_desc {
  const float* A;        int64_t lda;     /* row stride of A in elements                       */
  const int32_t* a_idx;  int32_t a_rows;  /* rows per clip in A (N when gathered, kcap compact) */
  const float* W;        const float* bias;
  float* out;            int64_t ldo;
  const int32_t* o_idx;  int32_t o_rows;
  const int32_t* count;                    /* nullable: per-clip valid rows                     */
  float* p_upd;                            /* nullable: gate reference to refresh (ld = lda)    */
  int32_t B, kcap, K, Nout;
  int32_t act;                             /* evt_act                                           */
  const void* W_split;                     /* nullable: evt_split_weights(W) -> split-precision  */
                                           /* MFMA path (3 bf16 MFMAs per fp32 product)          */
  void* workspace;       int64_t workspace_bytes;  /* nullable: split-K partial sums (see below) */
  int32_t a_bf16;                          /* ABI 4.  1: A points to bf16 values (row stride lda elements) -- activations that */
                                           /* are exactly bf16-representable, i.e. the A.v state of a bf16 matmul_2_cast,    */
                                           /* which IS the attention output (blocks.py:183-189): half the bytes, no split,   */
                                           /* one MFMA of three skipped, bit-identical results.  p_upd still receives fp32.  */
                                           /* Only launches for which evt_gated_linear_big_tile() != 0 accept it.            */
} evt_linear_desc;

EVT_API int evt_gated_linear(const evt_linear_desc* d, void* stream);

/* The tile configuration of the persistent 256-row kernel this launch would run on (0: it runs on the 128x128 kernel or its
 * split-K form).  Shape-only: pointers are compared with NULL, nothing is dereferenced, no GPU needed. */
EVT_API int evt_gated_linear_big_tile(const evt_linear_desc* d);

/* Split-K for launches with few output tiles (one stream, small r: ViTDet, batch 1).  When the
 * 128x128 tiling yields fewer workgroups than the chip has CUs, the split-precision kernel divides K
 * among S workgroups per tile; each writes its fp32 partial tile to `workspace` and a second kernel
 * sums the S partials IN FIXED ORDER (bit-reproducible), adds bias, applies act and scatters through
 * o_idx.  Without `count`, S depends only on (B*kcap, K, Nout).  With `count` (threshold policy:
 * kcap = N but few live rows) and B <= 32, S is chosen ON THE DEVICE from the counts -- the launch
 * covers the largest S the shape allows, dead tiles and surplus splits exit at once -- so S is a
 * function of the data, still identical on every rerun.  This returns the bytes K3 wants for a
 * given shape (0 = split-K not used); a smaller or null workspace silently selects the single-pass
 * kernel. */
EVT_API int64_t evt_gated_linear_workspace_bytes(int32_t B, int32_t kcap, int32_t K, int32_t Nout, int32_t has_count);

/* Split an fp32 weight matrix (rows, cols) row-major into bf16 hi/lo planes, hi = rne_bf16(w), lo = rne_bf16(w - hi),
 * in the "hl32" layout the GEMM kernels stream: every row is ceil(cols / 32) groups of 128 bytes
 * [32 x bf16 hi | 32 x bf16 lo] of 32 consecutive columns (zero past `cols`), so one 32-wide k-tile of a row is one
 * aligned 128-byte line holding both planes.  `out` must hold evt_split_weights_bytes(rows, cols) bytes.
 * With W_split set, K3/K7 compute a.w = a_hi.w_hi + a_hi.w_lo + a_lo.w_hi on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (activations are split on the fly, in registers); the dropped a_lo.w_lo term is <= 2^-16 per product,
 * i.e. fp32 activations still agree with the reference to ~1e-5 relative (tolerance 1e-3), at ~5x the fp32-input
 * MFMA rate.  W_split == NULL keeps the exact fp32 MFMA kernel.  (The reference keeps fp32 weights only:
 * counting.py:127-162.) */
EVT_API int evt_split_weights(const float* w, void* out, int64_t rows, int64_t cols, void* stream);
EVT_API int64_t evt_split_weights_bytes(int64_t rows, int64_t cols);

/* K7  Gated MLP: hidden = GELU(A[rows].W1^T + b1) -> out[rows] = hidden.W2^T + b2, as two MFMA
 * launches sharing one compact `hidden` scratch (B*kcap, Dh) provided by the caller.
 * Replaces Block._forward_mlp on gated rows + mlp_accumulator scatter (blocks.py:242-246, 446-447). */
typedef struct evt_mlp_desc {
  const float* A;        int64_t lda;
  const int32_t* idx;    int32_t rows;    /* idx gathers A rows and scatters out rows; N per clip */
  const float* W1; const float* b1; const float* W2; const float* b2;
  float* hidden;                          /* (B*kcap, Dh) scratch                               */
  float* out;            int64_t ldo;
  const int32_t* count;
  float* p_upd;
  int32_t B, kcap, D, Dh;
  const void* W1_split; const void* W2_split;   /* nullable pair: split-precision MFMA path      */
  void* workspace;       int64_t workspace_bytes; /* nullable: split-K partials, max over both   */
                                                  /* linears of evt_gated_linear_workspace_bytes  */
} evt_mlp_desc;

EVT_API int evt_gated_mlp(const evt_mlp_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K4  q.k^T product state: product (G, H, Nq, Nk) fp32, q divided by `scale` (= sqrt(D/H),
 *     blocks.py:92,514) before the contraction, fp32 MFMA.
 *   delta == 0: product = (q/scale) k^T                         (MatmulBuffer.forward_first,
 *               modules.py:224-230)
 *   delta == 1: rows idx_q <- (q/scale)[idx_q] k^T, then columns idx_k <- (q/scale) k[idx_k]^T,
 *               both from the already-updated operands (modules.py:232-248) in ONE launch.
 *   Addressing: element (clip c, head h, token t, channel d) of q is q[c*q_bs + t*q_rs + h*q_hs + d]
 *   (same for k), so the packed (B,N,3D) token buffer of the blocks (q = buf, k = buf + D,
 *   rs = 3D, hs = dh, bs = N*3D; blocks.py:248-255) and free-standing (B,H,N,dh) tensors both fit.
 *   tok_map (nullable, (groups_per_clip, Nq)): window partition for windowed attention
 *     (blocks.py:257-301), identical for every clip: token t of group g is token
 *     tok_map[(g % groups_per_clip)*Nq + t] of clip g / groups_per_clip, or the padding vector
 *     pad_q / pad_k (slices of the qkv bias, blocks.py:280-281) when negative.  With a map, G counts
 *     groups (clips x windows), Nq == Nk == window length, delta must be 0.
 * dh % 8 == 0, dh <= 128; strides multiples of 4 elements.
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_qk_desc {
  const float* q; int64_t q_bs, q_hs, q_rs;
  const float* k; int64_t k_bs, k_hs, k_rs;
  float* product;
  const int32_t* idx_q; const int32_t* count_q; int32_t kcap_q;   /* delta only                  */
  const int32_t* idx_q_rest;  /* nullable (G, Nq): complement of idx_q; the column panel then covers only these rows
                                 (the rows in idx_q are rewritten whole by the row panel) */
  const int32_t* idx_k; const int32_t* count_k; int32_t kcap_k;
  const int32_t* tok_map; int32_t groups_per_clip; const float* pad_q; const float* pad_k;
  int32_t G, H, Nq, Nk, dh;
  float scale;
  int32_t delta;
  int32_t split;   /* 0: fp32-input MFMA (exact fp32 products); 1: q, k split into bf16 hi/lo planes on the fly,
                      (q/scale).k = lo.hi + hi.lo + hi.hi on the bf16 MFMA, fp32 accumulate (~1e-5 relative) */
} evt_qk_desc;

EVT_API int evt_qk(const evt_qk_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K5  Row softmax over the product state with optional decomposed relative position terms,
 *     fused with the attention ("A") delta gate in column structure.
 *   logits[b,h,i,j] = product[b,h,i,j] + q_i.Ry[y_i, ky_j] + q_i.Rx[x_i, kx_j]   (utils.py:159-168,
 *                     q UNSCALED, blocks.py:521)
 *   a = softmax_j(logits)  (blocks.py:522), rounded to `store` (blocks.py:561)
 *   mode FULL : a_state[b,h,i,:] = a                       (TokenDeltaGate.forward_first, modules.py:183-185)
 *   mode GATED: for jj < count: j = idx[jj]
 *                 a_new[b,h,i,jj]   = a[j]
 *                 a_delta[b,h,i,jj] = round(a[j] - a_state[b,h,i,j])
 *                 a_state[b,h,i,j]  = a[j]                  (modules.py:187-201, structure="col")
 *   rel_y/rel_x: (qh, gh, dh) / (qw, gw, dh) tables (utils.py:151-156) or NULL.  gh x gw is the KEY grid
 *   (gh*gw == Nk), qw the query grid width; they differ only with K/V pooling (utils.py:143-146,185-188).
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_softmax_desc {
  const float* product;                   /* (B,H,N,Nk)                                          */
  const float* qkv;                       /* (B,N,3D) for the rel-pos terms; nullable if no rel  */
  const float* rel_y; const float* rel_x; int32_t gh, gw;   /* KEY grid (gh*gw == Nk)                */
  const int32_t* tok_map; int32_t groups_per_clip, clip_rows; const float* pad_row;
  void* a_state;                          /* (B,H,N,Nk) in `store` type                          */
  void* a_new; void* a_delta;             /* (B,H,N,kcap) in `store` type (GATED only)           */
  const int32_t* idx; const int32_t* count;
  int32_t B, H, N, Nk, D, kcap;
  int32_t store;                          /* evt_dtype                                           */
  int32_t gated;                          /* 0 = FULL, 1 = GATED                                 */
  int32_t qw;                             /* query grid width (== gw unless K/V are pooled)      */
} evt_softmax_desc;

EVT_API int evt_softmax_gate(const evt_softmax_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K6a Value delta gate (rows, forced index; blocks.py:566-567, modules.py:187-201):
 *   FULL : v_state[b,j,:] = round(v[b,j,:])  for all tokens
 *   GATED: v_new = round(v[idx]); v_delta = round(v_new - v_state[idx]);
 *          v_old = round(v_new - v_delta)   (the `v_n_tilde - v_delta_tilde` of modules.py:294)
 *          v_state[idx] = v_new
 *   v points at the value slice of token 0 (qkv + 2D of the packed buffer, or kv + D of the pooled
 *   buffer of evt_pool_kv), rows v_rs elements apart; pad_row likewise points at its value slice.  v_state: (B,N,D); v_delta, v_old: (B,kcap,D) with heads
 *   side by side (h*dh + d), or, with `transposed`, (B,H,dh,kcap) = (B,D,kcap) with k contiguous
 *   (the operand layout of evt_softmax_av_gated; columns [count, kcap) are written as zeros); all
 *   in `store` type.
 * ------------------------------------------------------------------------------------------ */
EVT_API int evt_v_gate(const float* v, int64_t v_rs, const int32_t* idx, const int32_t* count, int B, int N,
               int D, int kcap, void* v_state, void* v_delta, void* v_old, int store, int gated,
               int transposed, const int32_t* tok_map, int groups_per_clip, int clip_rows,
               const float* pad_row, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K6  Attention-value product state, heads merged on write (blocks.py:328-344 fused):
 *   FULL : pv[b,i,h*dh+d] = round( sum_j a_state[b,h,i,j] v_state[b,j,h*dh+d] )
 *          (MatmulDeltaAccumulator.forward_first, modules.py:277-283)
 *   GATED: pv += round( a_new . v_delta );  pv += round( a_delta . v_old )   each `+=` rounded
 *          to `store` (modules.py:285-295)
 *   out_f32 (nullable): pv converted back to fp32 (blocks.py:393-396), (B,N,D).
 *   out_map (nullable): un-window on write (blocks.py:346-376): row (g,t) -> clip row
 *   tok_map[g*N+t]; rows mapped to padding are dropped.
 *   Head dim D / H: any multiple of 16 up to 128 (16 .. 128; ViT-H's 80 included).
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_av_desc {
  const void* a1; const void* v1;         /* FULL: a_state (B,H,N,Nk), v_state (B,Nk,D)          */
  const void* a2; const void* v2;         /* GATED second product (a_delta, v_old); FULL: NULL   */
  int64_t lda;                            /* row stride of a1/a2 (Nk or kcap)                    */
  const int32_t* count;                   /* GATED: valid K per clip                             */
  void* pv;                               /* (B,N,D) in `store` type; nullable when FULL+out_f32 */
  float* out_f32;
  const int32_t* out_map; int32_t groups_per_clip, clip_rows;
  int32_t B, H, N, K, D;
  int32_t store;
  int32_t gated;
} evt_av_desc;

EVT_API int evt_av(const evt_av_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K5+K6 fused for gated frames (EventfulBlock._forward_attention, blocks.py:558-575, t >= 1):
 * softmax statistics of each state row (+ rel-pos), gather of the selected columns with the
 * attention delta gate (a~, da~, reference update), and both accumulator products on the matrix
 * cores, in one launch -- a~ / da~ never reach HBM.  Same results as evt_softmax_gate(gated) +
 * evt_av(gated).  v_delta_t / v_old_t: (B,D,kcap) = (B,H,dh,kcap) from evt_v_gate(transposed=1).
 * dh must be 64 or 128 (else use the two-kernel path).  bf16/fp16 store types run on
 * v_mfma_f32_32x32x16_{bf16,f16} (operands are exactly bf16/fp16 values, fp32 accumulate);
 * fp32 on v_mfma_f32_32x32x2_f32.
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_softmax_av_desc {
  const float* product;                   /* (B,H,N,Nk) q.k^T state; NULL: the scores are computed  */
                                          /* in the kernel as (q/scale).k^T from `qkv` (head dim 64, */
                                          /* N == Nk <= 256, kcap > 0) -- no state read, no evt_qk   */
  const float* qkv;                       /* (B,N,3D); read for rel-pos and when product == NULL     */
  const float* rel_y; const float* rel_x; int32_t gh, gw;
  void* a_state;                          /* (B,H,N,Nk) store type: matmul_gate.p                */
  const int32_t* idx; const int32_t* count; int32_t kcap;
  const void* v_delta_t; const void* v_old_t;
  void* pv;                               /* (B,N,D) store type: matmul_accumulator_2.product    */
  float* out_f32;                         /* (B,N,D); ABI 4: nullable with a 16-bit store type   */
                                          /* (the output then equals the pv state, widened)       */
  int32_t B, H, N, D, dh;
  int32_t store;
  int32_t Nk, qw;                         /* key count (== N unless pooled), query grid width    */
  float scale;                            /* product == NULL: q / scale (blocks.py:514)          */
  int32_t qk_split;                       /* product == NULL: 1 = q, k as bf16 hi + lo (3 bf16 MFMAs */
                                          /* per product, like evt_qk split), 0 = exact fp32 MFMA    */
  const float* norm_ref;                  /* nullable (B,N,D): the NEXT gate's reference (projection_gate.p) */
  float* norm_parts;                      /* with norm_ref: (B,N,H) out, ||out_f32 - norm_ref||^2 over the   */
                                          /* head's channels -> evt_select_*_sq(parts = H)                   */
  const float* rel_terms;                 /* ABI 4, nullable, with rel_y/rel_x: (B,H,N,gh+gw) from           */
                                          /* evt_rel_terms -- the per-row rel-pos dot products are read      */
                                          /* instead of recomputed by every 32-row workgroup                 */
} evt_softmax_av_desc;

EVT_API int evt_softmax_av_gated(const evt_softmax_av_desc* d, void* stream);

/* Decomposed relative position terms of every query token (utils.py:159-168, the two einsums of
 * `add_decomposed_rel_pos`): terms[b,h,i,e] = q[b,i,h,:] . rel_y[i / qw, e, :] for e < gh and
 * q[b,i,h,:] . rel_x[i % qw, e - gh, :] for gh <= e < gh + gw.  One workgroup per (clip, head, query-grid row or
 * column): its rel_y / rel_x slice and its q rows are read once (a 32-row attention workgroup computing the same
 * terms for itself re-reads 32 x (gh + gw) table rows of 256 bytes).  Head dim 64, N = qh * qw.
 * split (ABI 5): 1 = on the matrix cores with q and the tables as bf16 hi + lo (three bf16 MFMAs per product, ~1e-5
 * relative: the arithmetic of split-mode scores), 0 = fp32 FMA chains. */
EVT_API int evt_rel_terms(const float* qkv, const float* rel_y, const float* rel_x, int32_t B, int32_t H, int32_t N,
                          int32_t D, int32_t gh, int32_t gw, int32_t qw, int32_t split, float* terms, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K5+K6 for LARGE token counts, scores computed in the kernel (ABI 5; ViTDet global blocks, N = 1764 / 4096):
 * ONE launch per frame and block replaces evt_qk + evt_softmax_av_gated (gated frames) and evt_qk +
 * evt_softmax_gate + evt_av (first frame), and the (B,H,N,N) fp32 q.k^T state is neither kept nor read.
 *
 *   scores   x[i,j] = (q[i] / scale) . k[j]  (+ rel_terms[i, j / gw] + rel_terms[i, gh + j % gw])   blocks.py:514-521
 *            recomputed on the matrix cores from the packed token buffer for the 32 rows of a workgroup against all N
 *            keys; only running (max, sum) pairs are kept (softmax, blocks.py:522)
 *   first=1  a = round(softmax(x)) -> a_state_t (the matmul_gate reference, modules.py:183-185);
 *            pv = out = round(a . v_state)  (MatmulDeltaAccumulator first frame, modules.py:277-283)
 *   first=0  for the selected key columns idx: a~ = round(softmax(x)[:, idx]); da~ = round(a~ - ref[:, idx]);
 *            ref[:, idx] = a~  (TokenDeltaGate "col", modules.py:187-201);
 *            pv += round(a~ . dv~); pv += round(da~ . v_old)  (modules.py:285-295); out = pv, heads merged.
 *
 * a_state_t is the gate reference TRANSPOSED, (B,H,Nkeys,Nrows): the 32 rows a workgroup owns of a selected column
 * are contiguous.  Head dim 64, un-pooled keys (Nk == N).  v_delta_t / v_old_t, norm_ref / norm_parts,
 * out_f32 == NULL: as in evt_softmax_av_desc.  rel_terms: evt_rel_terms output or NULL (no relative position).
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_attn_stream_desc {
  const float* qkv;                       /* (B,N,3D) packed token buffer (qkv_accumulator.b)                */
  const float* rel_terms; int32_t gh, gw; /* nullable (B,H,N,gh+gw); key grid gh x gw == N                   */
  void* a_state_t;                        /* (B,H,N,N) store type, [b][h][key][row]: matmul_gate.p^T         */
  const int32_t* idx; const int32_t* count; int32_t kcap;   /* gated frame: selected key columns             */
  const void* v_delta_t; const void* v_old_t;               /* gated frame: (B,D,kcap) from evt_v_gate       */
  const void* v_state;                    /* first frame: (B,N,D) store type, round(v) (evt_v_gate FULL)     */
  void* pv;                               /* (B,N,D) store type: matmul_accumulator_2.product                */
  float* out_f32;                         /* (B,N,D); nullable on gated frames with a 16-bit store type      */
  const float* norm_ref; float* norm_parts;   /* gated frame, nullable pair: see evt_softmax_av_desc         */
  int32_t B, H, N, D;
  int32_t store;                          /* evt_dtype of a_state_t / v_* / pv                               */
  float scale;                            /* q / scale (blocks.py:514)                                       */
  int32_t qk_split;                       /* 1 = q, k as bf16 hi + lo (3 bf16 MFMAs per product), 0 = exact fp32 MFMA */
  int32_t first;                          /* 1 = first frame of a clip                                       */
  void* k_split;                          /* workspace, B * H * evt_attention_stream_key_blocks(N, gh, gw) * 4096 bytes, required
                                             with qk_split: the frame's key rows as bf16 hi / lo MFMA fragments (written by a
                                             pre-kernel of this call) */
  int32_t k_split_ready;                  /* ABI 6: 1 = evt_stream_prep has already written k_split for this frame: no pre-kernel */
  /* ABI 8: pooled keys / values (`pool_size`, blocks.py:303-326, 509-511).  kv = the (B,Nk,2D) buffer of evt_pool_kv (keys | values
   * per row), Nk = gh * gw pooled cells: the keys are its rows instead of the packed buffer's, a_state_t is (B,H,Nk,N), v_state
   * (B,Nk,D), idx / count / kcap the POOLED index list of evt_pool_index (blocks.py:525-540), v_delta_t / v_old_t the value gate of
   * kv's value half.  rel_terms (B,H,N,gh+gw) are the query tokens' terms against the pooled key grid.  NULL / 0: un-pooled
   * (Nk = N).  k_split then holds evt_attention_stream_key_blocks(Nk, gh, gw) blocks. */
  const float* kv; int32_t Nk;
} evt_attn_stream_desc;

/* Limits: head dim 64 (D == 64 H), N <= 32767 (32-bit byte offsets into a head's N x N reference), the tile's LDS within a CU
 * (evt_attention_stream_lds_bytes); EVT_ERR_BAD_SHAPE otherwise. */
EVT_API int evt_attention_stream(const evt_attn_stream_desc* d, void* stream);

/* LDS bytes of evt_attention_stream's smallest (32-row) tile for a store type and rel-pos key grid (gh = gw = 0: no relative
 * position); negative for an unknown store type.  Shape-only: callable without a device.  A caller routes a shape whose
 * answer exceeds the CU's 160 KB to evt_qk + evt_softmax_av_gated (the reference ops are the same, blocks.py:506-523,558-575);
 * evt_attention_stream itself returns EVT_ERR_BAD_SHAPE for it. */
EVT_API int64_t evt_attention_stream_lds_bytes(int32_t store, int32_t gh, int32_t gw);

/* 16-key blocks (4 KB per head each) of the k_split workspace: ceil(N / 16) without a rel-pos key grid (gh = gw = 0);
 * gh * ceil(gw / 16) with one -- every grid row starts a new block, so that a block's rel-pos terms are one row term and 16
 * consecutive column terms (utils.py:159-172).  Shape-only; negative for N <= 0. */
EVT_API int64_t evt_attention_stream_key_blocks(int32_t N, int32_t gh, int32_t gw);

/* ABI 6.  The three preparations of a GATED frame of evt_attention_stream in ONE launch (a role per workgroup range): the rel-pos
 * terms of every query token (what evt_rel_terms writes with split = 1: utils.py:159-168), the key plane `k_split` (then pass
 * k_split_ready = 1 to evt_attention_stream) and the value delta gate with transposed outputs (what evt_v_gate writes with
 * gated = transposed = 1 on the value slice of the packed buffer: modules.py:187-201 + blocks.py:561-567).  All three read the
 * updated token buffer and none reads another's output.  Same kernels' bodies: same results bit for bit.  Head dim 64,
 * N == gh * gw == qh * qw, kcap a positive multiple of 8.  A one-stream frame runs this once per global block instead of three
 * launches. */
typedef struct evt_stream_prep_desc {
  const float* qkv;                        /* (B,N,3D) packed token buffer (qkv_accumulator.b)                              */
  const float* rel_y; const float* rel_x;  /* (qh,gh,64), (qw,gw,64) rel-pos tables                                          */
  float* terms;                            /* out (B,H,N,gh+gw)                                                              */
  void* k_split;                           /* out: evt_attn_stream_desc.k_split                                              */
  const int32_t* idx; const int32_t* count; int32_t kcap;   /* the qkv gate's selected tokens (count nullable)               */
  void* v_state;                           /* in/out (B,N,D) store type: v_gate.p                                            */
  void* v_delta_t; void* v_old_t;          /* out (B,D,kcap) store type                                                      */
  int32_t B, H, N, D, gh, gw, qw;
  int32_t store;                           /* evt_dtype of v_state / v_delta_t / v_old_t                                     */
  /* ABI 8: pooled keys / values, as in evt_attn_stream_desc: kv = the (B,Nk,2D) buffer of evt_pool_kv, Nk = gh * gw; the key
   * plane and the value gate then read kv's rows (v_state (B,Nk,D); idx / count / kcap the pooled list of evt_pool_index), the
   * rel-pos terms stay those of the N query tokens.  NULL / 0: un-pooled (gh * gw == N). */
  const float* kv; int32_t Nk;
} evt_stream_prep_desc;

EVT_API int evt_stream_prep(const evt_stream_prep_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K/V token pooling (`pool_size`, blocks.py:303-326, 525-540).
 * evt_pool_kv   : kv[b, (ky,kx), :] = mean over the p0 x p1 cell of [k | v] of the packed (B, qh*qw, 3D)
 *                 token buffer -> (B, (qh/p0)*(qw/p1), 2D).  Replaces the avg_pool2d of `_pool_tokens`.
 * evt_pool_index: selected tokens -> pooled cells, de-duplicated, ascending, per clip, with the count on
 *                 the device (`_pool_index`; the reference's `.unique(dim=-1)` equals this for batch 1,
 *                 the only batch size its pooled configs use).  idx_k: (B, kcap_k), count_k: (B,).
 * ------------------------------------------------------------------------------------------ */
EVT_API int evt_pool_kv(const float* qkv, int B, int qh, int qw, int D, int p0, int p1, float* kv, void* stream);
EVT_API int evt_pool_index(const int32_t* idx, const int32_t* count, int B, int kcap, int qw, int p0, int p1,
                           int kw, int Nk, int kcap_k, int32_t* idx_k, int32_t* count_k, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * K8  Dense attention of one group (clip, or window of a partitioned clip) in ONE launch:
 *       out[b, t, h*64+d] = round( round(softmax((q/scale) k^T + rel-pos)) . round(v) )
 *     for groups of at most 256 tokens and head dim 64 (ViT-B: 14x14 windows, ViViT's 197 tokens).
 *     Replaces Block._forward_attention (blocks.py:205-240) incl. window partition / un-partition
 *     (blocks.py:257-301, 346-376: tok_map as in evt_qk, rows mapped to padding are dropped on
 *     write) -- i.e. evt_qk + evt_softmax_gate + evt_v_gate + evt_av without the (G,H,N,N) score
 *     and probability tensors ever reaching HBM.  Same arithmetic and rounding points as those.
 *     Optional state outputs for the first frame of an EventfulBlock clip (un-windowed only):
 *       product (G,H,N,N) fp32 = (q/scale) k^T   MatmulBuffer.forward_first        modules.py:224-230
 *       a_state (G,H,N,N) store type = P          matmul_gate reference             modules.py:183-185
 *       pv      (G,N,D)   store type = out        MatmulDeltaAccumulator first      modules.py:277-283
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_attn_dense_desc {
  const float* qkv;                       /* packed (B, clip_rows, 3D): q | k | v per token row   */
  const float* rel_y; const float* rel_x; /* nullable pair: (qh, gh, 64), (qw, gw, 64) tables     */
  int32_t gh, gw, qw;                     /* key grid gh x gw (= N), query grid width             */
  const int32_t* tok_map;                 /* nullable: (groups_per_clip, N) window map, -1 = pad  */
  int32_t groups_per_clip, clip_rows;
  const float* pad_row;                   /* qkv row of a padding token (the qkv bias)            */
  float* out_f32;                         /* (B, clip_rows, D)                                    */
  float* product; void* a_state; void* pv;/* nullable state outputs                               */
  int32_t G, H, N, D;                     /* G groups of N tokens                                 */
  float scale;
  int32_t store;
  int32_t qk_split;                       /* ABI 5: 1 = q, k and the rel-pos tables as bf16 hi + lo on the matrix cores (3 bf16  */
                                          /* MFMAs per product, ~1e-5 relative, like evt_qk split); 0 = exact fp32 products      */
  const float* norm_ref; float* norm_parts;   /* ABI 6, nullable pair: norm_ref (B, clip_rows, D) = the reference of the gate that  */
                                          /* consumes `out` (projection_gate.p); norm_parts (B, clip_rows, H) receives, per token   */
                                          /* and head, sum over the head's channels of (out - ref)^2 -- evt_select_*_sq then        */
                                          /* selects on sqrt of their sum (`c - self.p` + vector_norm, modules.py:149,               */
                                          /* policies.py:63/:28) and no separate pass reads the attention output.  Launches that    */
                                          /* run the resident kernel only: ask evt_attention_dense_resident first                   */
} evt_attn_dense_desc;

EVT_API int evt_attention_dense(const evt_attn_dense_desc* d, void* stream);
/* 1 when a launch of this shape WITHOUT state outputs runs the resident kernel (one workgroup per (group, head), the group's K / V
 * planes staged once: they must fit a CU's LDS); gh = gw = 0: no relative position.  Shape-only. */
EVT_API int evt_attention_dense_resident(int32_t N, int32_t gh, int32_t gw, int32_t store, int32_t qk_split);


/* ------------------------------------------------------------------------------------------ *
 * K10 (ABI 9).  The attention of EventfulBlock for AT MOST 256 TOKENS (every ViViT frame), resident form: one workgroup per
 * (clip, head), first frame or gated frame in ONE launch, the value delta gate included.  Replaces evt_v_gate +
 * evt_softmax_av_gated (gated frames) and evt_attention_dense(state outputs) + evt_v_gate (first frame) where it applies:
 *
 *   first=0  v~ = round(v[idx]); dv~ = round(v~ - v_state[idx]); v_old = round(v~ - dv~); v_state[idx] = v~
 *                                                               (TokenDeltaGate rows, modules.py:187-201; blocks.py:561-567)
 *            x = (q / scale) k^T from the packed token buffer, softmax over the keys              (blocks.py:514-522)
 *            a~ = round(softmax(x))[:, idx]; da~ = round(a~ - ref[:, idx]); ref[:, idx] = a~       ("col" gate, modules.py:187-201)
 *            pv += round(a~ . dv~); pv += round(da~ . v_old); out = pv, heads merged              (modules.py:285-295)
 *   first=1  ref = round(softmax(x)); v_state = round(v); pv = out = round(ref . v_state)         (modules.py:183-185, 277-283)
 *
 * The head's keys (bf16 hi | lo planes) and value operands are staged once in LDS; the value operands are scattered into
 * full-length [channel][key] planes with zero columns for the keys that are not selected, so the accumulator products contract
 * over all keys and the probabilities never leave the matrix-core register layout.
 *
 * a_tiles is the gate reference (matmul_gate.p) in a TILED layout, evt_attention_gated_tile_bytes(B, H, N) bytes: per (clip, head)
 * NT x NT tiles (NT = ceil(N / 32)) of 32 query rows x 32 keys, tile (rt, kb) at ((b H + h) NT + rt) NT + kb, 2 KB each; element
 * (row r, key c) of a tile at 512 (c / 16) + 8 (32 (c / 4 % 2) + r) + 4 (c / 8 % 2) + c % 4.  Rows / keys past N are padding
 * (written, never meaningful).  The host exposes the logical (B,H,N,N) tensor (eventful_transformer/modules.py).
 * Head dim 64, 16-bit store type, qk_split = 1 (q, k as bf16 hi + lo, three bf16 MFMAs per product); idx ascending, no duplicates.
 * ------------------------------------------------------------------------------------------ */
typedef struct evt_attn_gated_desc {
  const float* qkv;                       /* (B,N,3D) packed token buffer (qkv_accumulator.b), selected rows already updated */
  void* a_tiles;                          /* tiled gate reference, store type (see above)                       */
  const int32_t* idx; const int32_t* count; int32_t kcap;   /* gated frame: selected keys (B,kcap), 0 < kcap <= N */
  void* v_state;                          /* (B,N,D) store type: v_gate.p, read-modify-written (first: written)  */
  void* pv;                               /* (B,N,D) store type: matmul_accumulator_2.product                   */
  float* out_f32;                         /* (B,N,D), nullable: the output equals the pv state, widened          */
  const float* norm_ref; float* norm_parts;   /* nullable pair: see evt_softmax_av_desc                          */
  int32_t B, H, N, D;
  int32_t store;                          /* EVT_BF16 or EVT_F16                                                */
  float scale;                            /* q / scale (blocks.py:514)                                          */
  int32_t qk_split;                       /* must be 1                                                          */
  int32_t first;                          /* 1 = first frame of a clip                                          */
} evt_attn_gated_desc;

EVT_API int evt_attention_gated(const evt_attn_gated_desc* d, void* stream);
/* 1 when evt_attention_gated takes this shape / store type / score arithmetic.  Shape-only. */
EVT_API int evt_attention_gated_fits(int32_t N, int32_t D, int32_t H, int32_t store, int32_t qk_split);
/* Bytes of the tiled gate reference. */
EVT_API int64_t evt_attention_gated_tile_bytes(int32_t B, int32_t H, int32_t N);


/* ------------------------------------------------------------------------------------------ *
 * ABI 9.  The index-structured operations of the reference's modules that are NOT on the fused path -- stand-alone gates /
 * buffers of either structure and any element type, the window partition with padding, adaptive token sampling -- as kernels
 * (they ran on ATen gather / scatter / index kernels before).  `dtype` is an evt_dtype: the element type of the tensors.
 * ------------------------------------------------------------------------------------------ */
/* TokenGate / TokenDeltaGate with structure="col" (modules.py:154-164, 187-201): c, p are (Bp * R, N) -- Bp index rows, R tensor
 * rows per index row (e.g. Bp = B * H, R = N for an attention matrix) -- idx (Bp, kcap); c~ / e~ (nullable) are (Bp * R, kcap):
 * c~ = c[:, idx]; e~ = (c - p)[:, idx] in dtype arithmetic; update_p: p[:, idx] = c~. */
EVT_API int evt_gate_cols(const void* c, void* p, const int32_t* idx, const int32_t* count, int32_t Bp, int32_t R, int32_t N,
                          int32_t kcap, int32_t dtype, void* c_tilde, void* e_tilde, int32_t update_p, void* stream);
/* TokenBuffer with structure="col" (modules.py:90-96): buf[:, idx] = x, x (Bp * R, kcap), buf (Bp * R, N). */
EVT_API int evt_scatter_cols(const void* x, void* buf, const int32_t* idx, const int32_t* count, int32_t Bp, int32_t R, int32_t N,
                             int32_t kcap, int32_t dtype, void* stream);
/* Row-structured gate for any element type / row length (the fp32, F % 4 == 0 fast path is evt_gate_gather_update):
 * c, p (Bp, N, F); idx (Bp, kcap); c~ / e~ (Bp, kcap, F). */
EVT_API int evt_gate_rows_any(const void* c, void* p, const int32_t* idx, const int32_t* count, int32_t Bp, int32_t N, int32_t F,
                              int32_t kcap, int32_t dtype, void* c_tilde, void* e_tilde, int32_t update_p, void* stream);
/* Rows of a (B, N, F) tensor through an index map (B / rep, n), entries < 0 skipped: scatter = 0: out (B, n, F), out[b][i] =
 * x[b][map[b / rep][i]]; scatter = 1: x (B, n, F), out (B, N, F), out[b][map[b / rep][i]] = x[b][i] (TokenBuffer rows,
 * modules.py:86-88; the row gather of the ATS path, blocks.py:196-203).  Any element type and row length. */
EVT_API int evt_move_rows_any(const void* x, const int32_t* map, int32_t B, int32_t N, int32_t F, int32_t n, int32_t rep, int32_t scatter,
                              int32_t dtype, void* out, void* stream);
/* fp32 rows, F % 4 == 0, 16-byte pieces: out (B, n_out, F), out[b][i] = map[..][i] >= 0 ? x[b][map[..][i]] : pad_row (zeros when
 * NULL); map is (n_out) shared by the batch or, map_per_batch = 1, (B, n_out).  The window partition of the qkv buffer with its
 * padding tokens (blocks.py:257-301) and `_gather_ats_skip` (blocks.py:196-203). */
EVT_API int evt_gather_rows_map(const float* x, const int32_t* map, const float* pad_row, int32_t B, int32_t N, int32_t F, int32_t n_out,
                                int32_t map_per_batch, float* out, void* stream);
/* The inverse for un-windowing (blocks.py:346-376): x (B, n_in, F), out (B, N, F), out[b][map[i]] = x[b][i], map[i] < 0 dropped. */
EVT_API int evt_scatter_rows_map(const float* x, const int32_t* map, int32_t B, int32_t n_in, int32_t N, int32_t F, float* out, void* stream);

/* Adaptive token sampling, scoring (blocks.py:150-170): a (B,H,N,N) probabilities and v (B,H,N,dh) values (element strides v_bs,
 * v_hs, v_rs; channels contiguous), both of `dtype`; scores (H, N) fp32 = sum over the BATCH axis (blocks.py:163: batch must equal
 * heads downstream) of a[b,h,n,0] * ||v[b,h,n,:]|| / sum_{n' >= 1}(...), every intermediate rounded to `dtype` where the
 * reference's op sequence rounds it; scores[:, 0] = +inf.  Selection: evt_select_topk on the scores (ascending lists). */
EVT_API int evt_ats_scores(const void* a, const void* v, int64_t v_bs, int64_t v_hs, int64_t v_rs, int32_t B, int32_t H, int32_t N,
                           int32_t dh, int32_t dtype, float* scores, void* stream);
/* `_stabilize_ats_indices` (blocks.py:378-391): last, now (rows, n) ascending lists over N tokens -> out = last with the entries that
 * are not in `now` replaced, in order, by the entries of `now` that are not in `last`. */
EVT_API int evt_ats_stabilize(const int32_t* last, const int32_t* now, int32_t rows, int32_t n, int32_t N, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVT_ABI_H */
