"""Shared test helpers: build product (HIP) modules from oracle parameter dicts."""
import os

import numpy as np
import torch

import eventful_oracle as O

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The parity tests' measured numbers (agreement rates, worst errors, gate counts) are printed AND appended here, because
# `pytest -q` swallows stdout: scripts/collect_profiles.sh copies the file into profiles/rNN/parity_summary.txt.
PARITY_SUMMARY = os.environ.get("EVT_PARITY_SUMMARY", os.path.join(_ROOT, "gpurun_out", "parity_summary.txt"))


def report(line):
    """print + append to the parity summary file (best effort: a read-only tree must not fail a test)."""
    print(line)
    try:
        os.makedirs(os.path.dirname(PARITY_SUMMARY), exist_ok=True)
        with open(PARITY_SUMMARY, "a") as f:
            f.write(line.strip("\n") + "\n")
    except OSError:
        pass


def load_npz(path):
    return np.load(path, allow_pickle=False)


def product_block(kind, params, dim, heads, input_size, mlp_ratio=4, device="cuda", **kw):
    from eventful_transformer import blocks

    blk = getattr(blocks, kind)(dim=dim, heads=heads, input_size=input_size, mlp_ratio=mlp_ratio, **kw)
    res = blk.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return blk.eval().to(device)


def set_policies(model, policy_class, **kw):
    """utils/misc.py:140-143 of the reference, verbatim semantics: one fresh policy per gate."""
    from eventful_transformer.modules import SimpleSTGTGate, TokenDeltaGate, TokenGate

    for gate_class in [SimpleSTGTGate, TokenDeltaGate, TokenGate]:
        for gate in model.modules_of_type(gate_class):
            gate.policy = policy_class(**kw)


def oracle_policy(spec):
    if spec is None:
        return lambda: None
    if spec[0] == "topk":
        return lambda: O.TopK(spec[1])
    return lambda: O.Threshold(spec[1])


def product_policy(model, spec):
    from eventful_transformer import policies

    if spec is None:
        return
    if spec[0] == "topk":
        set_policies(model, policies.TokenNormTopK, k=spec[1])
    else:
        set_policies(model, policies.TokenNormThreshold, threshold=spec[1])


def sorted_set(index):
    return None if index is None else torch.sort(index.reshape(index.shape[0] if index.ndim > 1 else 1, -1).long(), dim=-1)[0]


# The small block cases of oracle/gen_golden.py (kept in sync by name through the fixture file).
SMALL = dict(dim=64, heads=4, mlp_ratio=4)


def small_cases():
    cs = {}
    for kind in ("EventfulTokenwiseBlock", "EventfulMatmul1Block", "EventfulBlock"):
        cs[f"{kind}_topk"] = (kind, (6, 6), True, {}, ("topk", 12))
    cs["EventfulBlock_bf16"] = ("EventfulBlock", (6, 6), True, dict(matmul_2_cast="bfloat16"), ("topk", 12))
    cs["EventfulBlock_fp16"] = ("EventfulBlock", (6, 6), True, dict(matmul_2_cast="float16"), ("topk", 12))
    cs["EventfulBlock_k_all"] = ("EventfulBlock", (6, 6), True, {}, ("topk", 37))
    cs["EventfulBlock_thr"] = ("EventfulBlock", (6, 6), False, {}, ("thr", 0.6))
    cs["EventfulBlock_thr_none"] = ("EventfulBlock", (6, 6), False, {}, ("thr", 1e9))
    cs["EventfulBlock_rel"] = ("EventfulBlock", (6, 6), False, dict(relative_embedding_size=(6, 6)), ("topk", 12))
    cs["EventfulBlock_rel_resized_bf16"] = ("EventfulBlock", (6, 6), False,
                                            dict(relative_embedding_size=(4, 4), matmul_2_cast="bfloat16"), ("topk", 12))
    cs["EventfulTokenwiseBlock_win"] = ("EventfulTokenwiseBlock", (6, 6), False,
                                        dict(window_size=(3, 3), relative_embedding_size=(8, 8)), ("topk", 12))
    cs["EventfulTokenwiseBlock_winpad"] = ("EventfulTokenwiseBlock", (7, 5), False,
                                           dict(window_size=(3, 3), relative_embedding_size=(8, 8)), ("topk", 12))
    cs["EventfulTokenwiseBlock_stgt"] = ("EventfulTokenwiseBlock", (6, 6), True, dict(stgt=True), ("topk", 12))
    cs["EventfulBlock_gate_before_ln"] = ("EventfulBlock", (6, 6), True, dict(gate_before_ln=True), ("topk", 12))
    cs["EventfulBlock_pool"] = ("EventfulBlock", (6, 6), False, dict(pool_size=2), ("topk", 12))
    cs["EventfulBlock_pool_rel_bf16"] = ("EventfulBlock", (6, 6), False,
                                         dict(pool_size=2, relative_embedding_size=(8, 8), matmul_2_cast="bfloat16"), ("topk", 12))
    cs["EventfulMatmul1Block_pool"] = ("EventfulMatmul1Block", (6, 6), False, dict(pool_size=(2, 3)), ("topk", 12))
    cs["EventfulBlock_pool_thr"] = ("EventfulBlock", (6, 6), False, dict(pool_size=2), ("thr", 0.6))
    cs["Block_pool_rel"] = ("Block", (6, 6), False, dict(pool_size=2, relative_embedding_size=(6, 6)), None)
    cs["Block_dense"] = ("Block", (6, 6), True, {}, None)
    cs["Block_win_rel"] = ("Block", (7, 5), False, dict(window_size=(3, 3), relative_embedding_size=(8, 8)), None)
    # K/V pooling INSIDE windows (blocks.py:308): no reference config uses it, the reference supports it
    cs["Block_winpool_rel"] = ("Block", (8, 8), False, dict(window_size=(4, 4), pool_size=2, relative_embedding_size=(8, 8)), None)
    cs["EventfulTokenwiseBlock_winpool_pad"] = ("EventfulTokenwiseBlock", (7, 6), False,
                                                dict(window_size=(4, 4), pool_size=2, relative_embedding_size=(8, 8)), ("topk", 12))
    return cs


def small_case_params(name, case, param_seed):
    kind, isz, has_cls, kw, pol = case
    rel = kw.get("relative_embedding_size")
    if rel is not None and kw.get("window_size"):
        rel = kw["window_size"]
    return O.make_block_params(SMALL["dim"], SMALL["mlp_ratio"], seed=int(param_seed), std=0.08, rel_sizes=rel,
                               head_dim=SMALL["dim"] // SMALL["heads"])


def backbone_params(depth, dim, mlp_ratio, seed, tokens, rel_for=None, std=0.02, qk_std=None):
    """Same generator as oracle/gen_golden.py::backbone_params (numpy RandomState => version-stable)."""
    rs = np.random.RandomState(seed)
    sd = {"position_encoding.encoding": torch.from_numpy((rs.standard_normal((1, tokens, dim)) * std).astype(np.float32))}
    for i in range(depth):
        rel = None if rel_for is None else rel_for(i)
        bp = O.make_block_params(dim, mlp_ratio, seed=seed * 100 + i, std=std, rel_sizes=rel, head_dim=64)
        if qk_std is not None:
            O.sharpen_qk(bp, dim, qk_std, std)
        for k, v in bp.items():
            sd[f"blocks.{i}.{k}"] = v
    return sd


def block_params_of(sd, i):
    pre = f"blocks.{i}."
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


def vivit_oracle(mode_cast, seed=77, k=128, qk_std=None, grid=14):
    """ViViT-B spatial oracle + its parameters, as in gen_golden.gen_vivit (qk_std: gen_vivit_sharp; grid 20: the EPIC-Kitchens model)."""
    dim, depth, heads, N = 768, 12, 12, grid * grid
    sd = backbone_params(depth, dim, 4, seed, N + 1, qk_std=qk_std)
    rs = np.random.RandomState(seed + 1)
    cls = torch.from_numpy((rs.standard_normal((1, 1, dim)) * 0.02).astype(np.float32))
    ln_w = torch.from_numpy((1 + rs.standard_normal(dim) * 0.05).astype(np.float32))
    ln_b = torch.from_numpy((rs.standard_normal(dim) * 0.05).astype(np.float32))
    blocks = [O.BlockOracle("EventfulBlock", block_params_of(sd, i), dim, heads, (grid, grid), matmul_2_cast=mode_cast)
              for i in range(depth)]
    ob = O.BackboneOracle(blocks, sd["position_encoding.encoding"], (grid, grid), (grid, grid), True)
    ob.set_policy(lambda: O.TopK(k))
    return O.ViViTSpatialOracle(ob, cls, ln_w, ln_b), sd, cls, ln_w, ln_b


VITDET_WINDOWED = (0, 1, 3, 4, 6, 7, 9, 10)  # configs/models/vitdet_b_coco.yml:13


def vitdet_oracle(grid, policy_factory, cast_global, seed, qk_std=None):
    dim, depth, heads = 768, 12, 12

    def rel_for(i):
        return (14, 14) if i in VITDET_WINDOWED else (64, 64)

    sd = backbone_params(depth, dim, 4, seed, 14 * 14, rel_for=rel_for, qk_std=qk_std)
    blocks = []
    for i in range(depth):
        if i in VITDET_WINDOWED:
            blocks.append(O.BlockOracle("EventfulTokenwiseBlock", block_params_of(sd, i), dim, heads, (grid, grid),
                                        window_size=(14, 14), relative_embedding_size=(64, 64)))
        else:
            blocks.append(O.BlockOracle("EventfulBlock", block_params_of(sd, i), dim, heads, (grid, grid),
                                        relative_embedding_size=(64, 64), matmul_2_cast=cast_global))
    ob = O.BackboneOracle(blocks, sd["position_encoding.encoding"], (14, 14), (grid, grid), False)
    ob.set_policy(policy_factory)
    return ob, sd


def product_vitdet(grid, sd, cast_global, device="cuda", pool_size=None):
    from eventful_transformer.backbones import ViTBackbone

    cfg = dict(dim=768, heads=12, mlp_ratio=4, relative_embedding_size=(64, 64), window_size=(14, 14))
    overrides = {}
    if cast_global:
        cfg["matmul_2_cast"] = cast_global
        overrides["matmul_2_cast"] = None
    if pool_size is not None:      # configs/evaluate/vitdet_vid/_spatial.yml:4-6: K / V pooling in the global blocks only
        cfg["pool_size"] = pool_size
        overrides["pool_size"] = None
    bb = ViTBackbone(block_config=cfg, depth=12, position_encoding_size=(14, 14), input_size=(grid, grid),
                     block_class="EventfulBlock", windowed_class="EventfulTokenwiseBlock",
                     window_indices=VITDET_WINDOWED,
                     windowed_overrides=(overrides or None))
    res = bb.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return bb.eval().to(device)


def product_vivit(sd, cast, device="cuda", grid=14):
    from eventful_transformer.backbones import ViTBackbone

    cfg = dict(dim=768, heads=12, mlp_ratio=4)
    if cast:
        cfg["matmul_2_cast"] = cast
    bb = ViTBackbone(block_config=cfg, depth=12, position_encoding_size=(grid, grid), input_size=(grid, grid),
                     block_class="EventfulBlock", has_class_token=True)
    res = bb.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return bb.eval().to(device)


def seeded_module_params(module, seed, std=0.02):
    """Same generator as oracle/gen_golden.py::seeded_module_params (sorted key order => independent of the
    registration order; numpy RandomState => version-stable)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, p in sorted(module.state_dict().items()):
        v = (rs.standard_normal(tuple(p.shape)) * std).astype(np.float32)
        if "layer_norm.weight" in name or (name.endswith(".weight") and p.ndim == 1):
            v = (1.0 + rs.standard_normal(tuple(p.shape)) * 0.05).astype(np.float32)
        sd[name] = torch.from_numpy(v)
    return sd


def synthetic_video(seed, frames=80):
    """(1, frames, 3, 224, 224) uint8 video of gen_golden.gen_models: 40 random 16x16 patches change per frame."""
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(1, 1, 3, 224, 224)).astype(np.uint8)
    out = [base[:, 0]]
    for t in range(1, frames):
        f = out[-1].copy()
        for _ in range(40):
            y, x = rs.randint(0, 14) * 16, rs.randint(0, 14) * 16
            f[:, :, y:y + 16, x:x + 16] = rs.randint(0, 256, size=(1, 3, 16, 16))
        out.append(f)
    return torch.from_numpy(np.stack(out, axis=1))


VIVIT_B_CONFIG = dict(   # configs/models/vivit_b_kinetics400.yml of the reference, with 1 spatial x 2 temporal views
    classes=400, input_shape=[32, 3, 224, 224], normalize_mean=0.45, normalize_std=0.225, spatial_views=1,
    temporal_stride=2, temporal_views=2, tubelet_shape=[2, 16, 16],
    spatial_config=dict(depth=12, position_encoding_size=[14, 14], block_config=dict(dim=768, heads=12, mlp_ratio=4),
                        block_class="EventfulBlock"),
    temporal_config=dict(depth=4, position_encoding_size=[16], block_config=dict(dim=768, heads=12, mlp_ratio=4)))
