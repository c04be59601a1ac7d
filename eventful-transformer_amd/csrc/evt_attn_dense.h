// evt_attn_dense.h -- argument block shared by the two kernels behind evt_attention_dense (K8):
// evt_attn_dense.hip (32-row tiles, optional state outputs) and evt_attn_window.hip (one workgroup per (group, head)).
#pragma once
#include "evt_common.h"

struct DenseArgs {
  const float* qkv; const float* rel_y; const float* rel_x;
  const int32_t* tok_map; const float* pad_row;
  float* out_f32; float* product; void* a_state; void* pv;
  int groups_per_clip, clip_rows;
  int G, H, N, D, gh, gw, qw;
  float scale;
  const float* norm_ref; float* norm_parts;   // resident kernel only: (rows, D) reference of the next gate -> (rows, H) partial ||out - ref||^2
};

// evt_attn_window.hip.  evt_window_fits: shape-only (LDS budget of the resident K / V planes); evt_launch_window returns false
// when it does not take the launch (state outputs requested, or the planes do not fit a CU's LDS).
bool evt_window_fits(int N, int nrel, int store, int split);   // (also exported as evt_attention_dense_resident)
bool evt_launch_window(const DenseArgs& a, int store, int split, hipStream_t s);
