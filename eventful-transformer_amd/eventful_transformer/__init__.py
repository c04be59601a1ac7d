"""MI355X-native implementation of the Eventful Transformer gated-token inference path.

Same package, module, class, kwarg, attribute and state_dict names as WISION-Lab/eventful-transformer's
`eventful_transformer` package, so its ViViT / ViTDet wrappers import and run unchanged; underneath,
the blocks drive hand-written gfx950 HIP kernels through the C ABI in include/evt_abi.h.
"""
