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

    def forward(self, x):
        x = self.position_encoding(x)
        x = self.blocks(x)
        return x
