"""CPU: properties of the ORACLE (oracle/eventful_oracle.py, the restatement of the reference's modules.py / blocks.py) that the GPU
property tests (tests/test_gpu_properties.py) rely on, at a size the CPU finishes in a second."""
import torch

import eventful_oracle as O


def test_oracle_constant_input_reaches_a_fixed_point_that_is_not_the_dense_pass():
    """EventfulBlock on a constant input (modules.py:122-201, 265-295; blocks.py:518-575): each of the three gates needs
    ceil(N / k) frames after its input settled, then every delta is exactly zero and the output stops moving bit for bit.  The fixed
    point is NOT the dense `Block`'s output: attention-gate columns refreshed while other tokens were still stale keep those
    probabilities until their key token is selected again -- the reference's approximation, not an implementation artefact (the GPU
    test reports the same gap at full size)."""
    dim, heads, n, k = 64, 4, 50, 20
    g = torch.Generator().manual_seed(0)
    for std, lo, hi in ((0.02, 1e-5, 5e-3), (0.1, 1e-3, 1.0)):
        params = O.make_block_params(dim, 4, seed=5, std=std)
        ev = O.BlockOracle("EventfulBlock", params, dim, heads, (7, 7))
        ev.set_policy(lambda: O.TopK(k))
        dense = O.BlockOracle("Block", params, dim, heads, (7, 7))
        xs = [torch.randn(1, n, dim, generator=g) for _ in range(3)]
        ev.reset()
        for x in xs:
            y = ev.forward(x).clone()
        frames = 0
        for frames in range(1, 3 * 3 + 3):
            y2 = ev.forward(xs[-1]).clone()
            same = torch.equal(y, y2)
            y = y2
            if same:
                break
        assert same, "the oracle's EventfulBlock keeps moving on a constant input"
        assert torch.equal(ev.forward(xs[-1]), y)
        dense.reset()
        yd = dense.forward(xs[-1])
        ev.reset()
        assert torch.equal(ev.forward(xs[-1]), yd)          # the first frame of a clip IS the dense pass
        gap = float((y - yd).abs().max())
        assert lo < gap < hi, (std, gap)
