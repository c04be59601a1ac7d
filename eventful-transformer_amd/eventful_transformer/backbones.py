"""ViT backbone (API of the reference's backbones.py): position encoding, then a stack of blocks
chosen BY CLASS NAME from `eventful_transformer.blocks` (backbones.py:46-59)."""

import torch.nn as nn

from eventful_transformer import blocks
from eventful_transformer.base import ExtendedModule
from eventful_transformer.utils import PositionEncoding


class ViTBackbone(ExtendedModule):
    """Common backbone for the ViViT sub-models and ViTDet (backbones.py:8-64)."""

    def __init__(self, block_config, depth, position_encoding_size, input_size, block_class="Block",
                 has_class_token=False, window_indices=(), windowed_class=None, windowed_overrides=None):
        super().__init__()
        self.position_encoding = PositionEncoding(block_config["dim"], position_encoding_size, input_size,
                                                  has_class_token)
        self.blocks = nn.Sequential()
        for i in range(depth):
            name, config = block_class, dict(block_config)
            if i in window_indices:
                if windowed_class is not None:
                    name = windowed_class
                if windowed_overrides is not None:
                    config.update(windowed_overrides)
            else:
                config["window_size"] = None
            self.blocks.append(getattr(blocks, name)(input_size=input_size, **config))
        # every block knows its successor (the first block follows the last: the next frame) -- its MLP gate's selection launch
        # prefetches the successor's QKV weight planes (blocks.py::_select).  Not a registered submodule.
        mods = list(self.blocks)
        for i, blk in enumerate(mods):
            object.__setattr__(blk, "_next_block", mods[(i + 1) % len(mods)])
        # graphs.FrameGraphs.run_pipelined sets this while it captures several frames of one stream side by side: an
        # object whose before_block(i, n) / after_block(i, n) order block i of a frame behind the previous frame's blocks
        self.block_sync = None

    def forward(self, x):
        x = self.position_encoding(x)
        # Between consecutive eventful blocks the residual stream travels as an unevaluated sum (blocks.PendingSum): the
        # next block's first row pass performs the add together with its LayerNorm + delta norm.  Same arithmetic, one
        # launch and one HBM round trip of the stream less per block boundary.  A PendingSum never reaches user code:
        # a block boundary is only chained when neither side (nor the container, nor torch's global registry) carries a
        # forward / pre-forward hook -- hooked blocks take and return plain tensors, and a hooked `blocks` container is
        # invoked as the nn.Sequential it is.
        sync = self.block_sync
        if (not CHAIN_BLOCKS or _hooked(self.blocks)) and sync is None:
            return self.blocks(x)
        chain = CHAIN_BLOCKS and not _hooked(self.blocks)
        mods = list(self.blocks)
        for i, blk in enumerate(mods):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if sync is not None:
                sync.before_block(i, len(mods))
            if chain and nxt is not None and _chains(blk) and _chains(nxt):
                x = blk(x, _defer_output=True)
            else:
                x = blk(x)
            if sync is not None:
                sync.after_block(i, len(mods))
        return x.materialize() if isinstance(x, blocks.PendingSum) else x


CHAIN_BLOCKS = True   # (module constant: block chaining through PendingSum; tests may turn it off to compare)


def _hooked(module):
    """True when a forward or pre-forward hook would observe this module's call (its own or a global one)."""
    from torch.nn.modules import module as _m

    return bool(module._forward_hooks or module._forward_pre_hooks or _m._global_forward_hooks
                or _m._global_forward_pre_hooks)


def _chains(blk):
    """Blocks that can pass / take a pending residual sum: un-hooked eventful classes without adaptive token sampling."""
    return isinstance(blk, blocks.EventfulTokenwiseBlock) and blk.ats_fraction is None and not _hooked(blk)
