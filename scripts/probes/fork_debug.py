import os, sys
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(os.getcwd(), p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
case, dh, heads, gh, gw, k, B = 31, 48, 1, 20, 2, 38, 1
dim = dh * heads; n = gh * gw
params = O.make_block_params(dim, 4, seed=case, std=0.05, head_dim=dh)
ob = O.BlockOracle("EventfulBlock", params, dim, heads, (gh, gw)); ob.set_policy(lambda: O.TopK(k))
blk = H.product_block("EventfulBlock", params, dim, heads, (gh, gw)); H.set_policies(blk, policies.TokenNormTopK, k=k)
xs = O.make_token_stream(B, n, dim, 4, k, seed=case + 1000, small=0.01)
class Forced(torch.nn.Module):
    def __init__(self, real):
        super().__init__(); self.real, self.force, self.mine, self.e = real, None, None, None
    def forward(self, e, dim=-1):
        self.mine = self.real(e, dim=dim); self.e = e.clone(); return self.force
gates = ("qkv_gate", "projection_gate", "mlp_gate")
for gn in gates: getattr(blk, gn).policy = Forced(getattr(blk, gn).policy)
with torch.inference_mode():
    for t in range(2):
        y_ref = ob.forward(xs[t])
        if t:
            for gn, tk in zip(gates, ("qkv_index", "projection_index", "mlp_index")):
                getattr(blk, gn).policy.force = ob.trace[tk].sort(dim=-1)[0].cuda()
        y = blk(xs[t].cuda()).cpu()
        if t:
            for gn, tk in zip(gates, ("qkv_index", "projection_index", "mlp_index")):
                pol = getattr(blk, gn).policy
                want = ob.trace[tk].sort(dim=-1)[0]
                mine = pol.mine.sort(dim=-1)[0].cpu()
                e_or = ob.policy[gn].last_input
                n_or = torch.linalg.vector_norm(e_or, dim=-1)[0]
                n_my = torch.linalg.vector_norm(pol.e.cpu(), dim=-1)[0]
                print(gn, "equal", torch.equal(want, mine), "oracle norms sorted desc around k:", n_or.sort(descending=True)[0][k-2:k+2].tolist(), "zeros:", int((n_or == 0).sum()),
                      "| product delta norms vs oracle's max diff", float((n_my - n_or).abs().max()), "want", want[0][:8].tolist(), "mine", mine[0][:8].tolist())
