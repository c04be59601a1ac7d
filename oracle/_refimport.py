"""Import the upstream reference (read-only mount at /root/reference) for golden-vector generation.

TEST INFRASTRUCTURE ONLY. Used by oracle/gen_golden.py in the build container; /root/reference does
not exist on the GPU box, and nothing in the product package, bench.py or the -m gpu tests imports
this module.

torchvision is absent from this image; the reference only needs `Normalize`, `InterpolationMode` and
module objects to exist at import time (utils/image.py:4-6, models/vivit.py:3), so a tiny stand-in is
injected into sys.modules before the import. matplotlib is stubbed the same
way if it is missing.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _install_stubs():
    import torch

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        trf = types.ModuleType("torchvision.transforms.functional")
        tio = types.ModuleType("torchvision.io")

        class Normalize(torch.nn.Module):
            def __init__(self, mean, std):
                super().__init__()
                self.mean, self.std = mean, std

            def forward(self, x):
                mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device)
                std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device)
                if mean.ndim:
                    mean = mean.view(-1, 1, 1)
                    std = std.view(-1, 1, 1)
                return (x - mean) / std

        class InterpolationMode:
            BILINEAR = "bilinear"
            BICUBIC = "bicubic"
            NEAREST = "nearest"

        tr.Normalize = Normalize
        tr.InterpolationMode = InterpolationMode
        tr.functional = trf
        tv.transforms = tr
        tv.io = tio
        sys.modules.update(
            {
                "torchvision": tv,
                "torchvision.transforms": tr,
                "torchvision.transforms.functional": trf,
                "torchvision.io": tio,
            }
        )
    if "detectron2" not in sys.modules:
        # models/vitdet.py imports `LazyConfig`, `instantiate` and `ImageList` at module level (vitdet.py:2-3); only the
        # detection heads (out of scope) use them.  Empty stand-ins let the file import so that its
        # LinearEmbedding / ViTDetPreprocessing / SimplePyramid classes can be run for golden vectors.
        d2 = types.ModuleType("detectron2")
        d2c = types.ModuleType("detectron2.config")
        d2s = types.ModuleType("detectron2.structures")
        d2c.LazyConfig = type("LazyConfig", (), {})
        d2c.instantiate = lambda *a, **k: None
        d2s.ImageList = type("ImageList", (), {})
        d2.config, d2.structures = d2c, d2s
        sys.modules.update({"detectron2": d2, "detectron2.config": d2c, "detectron2.structures": d2s})
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mp = types.ModuleType("matplotlib")
        mpp = types.ModuleType("matplotlib.pyplot")
        mp.pyplot = mpp
        sys.modules.update({"matplotlib": mp, "matplotlib.pyplot": mpp})


def import_reference():
    """Returns the reference's `eventful_transformer` package (and makes `models`, `utils` importable)."""
    sys.dont_write_bytecode = True  # keep the read-only mount pristine
    _install_stubs()
    # The product package has the same import name; make sure the reference wins in THIS process.
    for name in [m for m in sys.modules if m == "eventful_transformer" or m.startswith("eventful_transformer.")]:
        del sys.modules[name]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import eventful_transformer  # noqa: F401
    import eventful_transformer.blocks  # noqa: F401
    import eventful_transformer.backbones  # noqa: F401

    assert eventful_transformer.blocks.__file__.startswith(REFERENCE_ROOT)
    return eventful_transformer
