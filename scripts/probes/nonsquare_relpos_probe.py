import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
torch.set_num_threads(8)
for isz, cast in (((16, 18), None), ((16, 18), "float16"), ((12, 14), None), ((12, 14), "bfloat16"), ((30, 20), None)):
    n = isz[0] * isz[1]; k = n // 3
    kw = dict(relative_embedding_size=isz)
    if cast: kw["matmul_2_cast"] = cast
    try:
        params = O.make_block_params(768, 4, seed=n, std=0.02, rel_sizes=isz, head_dim=64)
        ob = O.BlockOracle("EventfulBlock", params, 768, 12, isz, **kw); ob.set_policy(lambda: O.TopK(k))
        blk = H.product_block("EventfulBlock", params, 768, 12, isz, **kw); H.set_policies(blk, policies.TokenNormTopK, k=k)
        xs = O.make_token_stream(1, n, 768, 3, k, seed=n + 1, small=0.01)
        with torch.inference_mode():
            errs = [float((blk(xs[t].cuda()).cpu() - ob.forward(xs[t])).abs().max()) for t in range(3)]
        print(isz, cast, ['%.0e' % e for e in errs], flush=True)
    except Exception as e:
        print(isz, cast, "RAISED", type(e).__name__, str(e)[:150], flush=True)
