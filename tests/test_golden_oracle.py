"""CPU: golden fixtures from the reference vs (host logic + CPU oracle).

Pins the oracle (and the host-side sampling / params / history logic) against
every committed golden vector.  Runs without a GPU.
"""
from __future__ import annotations

import pytest

from golden_cases import case_ids
from golden_cases import check_case
from golden_cases import load_cases
from parity_harness import use_engine

CASES = {case["name"]: case for case in load_cases()}


@pytest.mark.parametrize("name", case_ids())
def test_golden_case_on_oracle(name, oracle):
    with use_engine(oracle):
        check_case(CASES[name], "cpu")


def test_aten_known_answers_on_oracle(oracle):
    """the empirically pinned grid_sample semantics, frozen as known-answer vectors."""
    import known_answers

    known_answers.check(oracle, "cpu")


def test_aten_known_answers_on_the_installed_torch():
    """The same vectors against torch's own CPU grid_sample (re-verifies the pin on this box)."""
    import torch
    import torch.nn.functional as F

    import known_answers

    axis = torch.arange(5, dtype=torch.float32).view(1, 1, 1, 1, 5)
    x = torch.tensor(known_answers.COORDS)
    grid = torch.zeros(1, 1, 1, len(x), 3)
    grid[..., 0] = 2 * x / 4 - 1  # x is the fastest (W) axis of grid_sample
    nearest = F.grid_sample(axis, grid, mode="nearest", padding_mode="zeros", align_corners=True).reshape(-1)
    bilinear = F.grid_sample(axis, grid, mode="bilinear", padding_mode="zeros", align_corners=True).reshape(-1)
    mask = F.grid_sample(torch.ones_like(axis), grid, mode="bilinear", padding_mode="zeros", align_corners=True).reshape(-1)
    assert nearest.tolist() == known_answers.NEAREST
    torch.testing.assert_close(bilinear, torch.tensor(known_answers.BILINEAR), rtol=0, atol=2e-5)
    torch.testing.assert_close(mask, torch.tensor(known_answers.MASK), rtol=0, atol=2e-5)
    assert mask[4].item() == 0.5 and mask[5].item() == 0.5  # exactly one half: fails `mask > 0.5`
