"""Per-attribute quantization policy against a literal restatement of the reference's
debug driver (python/encode_3dgs_debug.py:326-381) -- host logic, no GPU needed."""
import numpy as np
import torch


def _reference_policy(Coeff, n_channels, budget=1024):
    attr_ranges = {'quats': (0, 4), 'scales': (4, 7), 'opacity': (7, 8), 'colors': (8, n_channels)}
    w = {'quats': 1.0 / 21.93, 'scales': 1.0 / 26.36, 'opacity': 1.0 / 42.22, 'colors': 1.0 / 38.67}
    total = sum(w.values())
    enc = torch.zeros_like(Coeff)
    steps = {}
    for name, (a, b) in attr_ranges.items():
        if a >= n_channels:
            continue
        blk = Coeff[:, a:b]
        rng = blk.max() - blk.min()
        levels = max(int(budget * w[name] / total), 2)
        step = max(rng / max(levels - 1, 1), 1e-6)
        step = step.item() if torch.is_tensor(step) else step
        steps[name] = (step, levels)
        enc[:, a:b] = torch.floor(Coeff[:, a:b] / step + 0.5)
    return enc, steps


def test_per_attribute_steps_match_the_debug_driver():
    from raht_3dgs_codec_amd.pipeline import per_attribute_steps
    rng = np.random.default_rng(3)
    for n_channels in (56, 11, 8, 7):
        Coeff = torch.from_numpy(rng.normal(size=(500, n_channels)) * rng.uniform(0.1, 30, size=(1, n_channels)))
        steps, table = per_attribute_steps(Coeff)
        enc_ref, ref = _reference_policy(Coeff, n_channels)
        assert steps.shape == (n_channels,)
        for name, (step, levels) in ref.items():
            assert table[name]["levels"] == levels
            assert abs(table[name]["step"] - step) <= 1e-12 * abs(step)
            a, b = table[name]["channels"]
            assert torch.all(steps[a:b] == table[name]["step"])
        assert sum(t["levels"] for t in table.values()) <= 1024
        enc = torch.floor(Coeff / steps.to(Coeff.dtype) + 0.5)
        assert torch.equal(enc, enc_ref)                  # same float64 steps, same division: the driver's integers


def test_zero_range_attribute_gets_the_floor_step():
    from raht_3dgs_codec_amd.pipeline import per_attribute_steps
    Coeff = torch.zeros((10, 12), dtype=torch.float64)
    Coeff[:, 8:] = torch.arange(40, dtype=torch.float64).reshape(10, 4)
    steps, table = per_attribute_steps(Coeff)
    assert table["quats"]["step"] == 1e-6 and table["opacity"]["step"] == 1e-6
    assert table["colors"]["step"] > 0.1
