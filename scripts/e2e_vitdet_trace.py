"""ViTDet 672^2 end to end (uint8 frame -> pyramid features), six eager frames of one stream: the program to put behind
`rocprofv3 --kernel-trace --stats --` to see what the pre- / post-backbone stages launch (the pyramid's GEMMs run on the
128x128 split-K kernel: 25 launches + 21 finish passes per frame, ~0.8 ms)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "eventful-transformer_amd"))
import numpy as np, torch
import bench
from eventful_transformer import policies
from models.vitdet import ViTDet
dev = torch.device("cuda", 0)
bcfg = dict(block_config=dict(dim=768, heads=12, mlp_ratio=4, relative_embedding_size=(64, 64), window_size=(14, 14)),
            depth=12, position_encoding_size=(14, 14), block_class="EventfulBlock", windowed_class="EventfulTokenwiseBlock",
            window_indices=bench.VITDET_WINDOWED)
det = ViTDet(bcfg, (3, 672, 672), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], 256, (16, 16), [4.0, 2.0, 1.0, 0.5])
det = det.eval().to(dev)
bench.set_policies(det, lambda: policies.TokenNormTopK(k=256))
g = torch.Generator(device=dev).manual_seed(11)
frames = torch.randint(0, 256, (6, 1, 3, 672, 672), dtype=torch.uint8, device=dev, generator=g)
with torch.inference_mode():
    det.reset()
    for t in range(6):
        images, x = det.pre_backbone(frames[t]); x = det.backbone(x)
        torch.cuda.synchronize()
        det.post_backbone(images, x)
        torch.cuda.synchronize()
