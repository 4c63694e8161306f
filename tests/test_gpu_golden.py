"""GPU: golden fixtures from the reference vs the HIP engine, through the public transform API."""
from __future__ import annotations

import pytest

from golden_cases import case_ids
from golden_cases import check_case
from golden_cases import load_cases

pytestmark = pytest.mark.gpu

CASES = {case["name"]: case for case in load_cases()}


@pytest.mark.parametrize("name", case_ids())
def test_golden_case_on_hip(name, hip):
    check_case(CASES[name], "cuda")


def test_aten_known_answers_on_hip(hip):
    """the empirically pinned grid_sample semantics, frozen as known-answer vectors."""
    import known_answers

    known_answers.check(hip, "cuda")


@pytest.mark.parametrize("name", case_ids())
def test_golden_case_on_the_lean_exact_kernel(name, hip, monkeypatch):
    """Round 5: the same fixtures — outputs of the unmodified reference — with the lean exact-coordinate kernel
    (csrc/resample_lean_exact.hpp) forced onto every launch it can take, whatever its size (`TIO_EXACT_LEAN=2`; by default
    only launches of >= 12 288 bricks take it, and the fixtures are small): anisotropic spacings, oblique affines, target
    grids, gated elements, fills, both composition orders go through its ATen-order instantiation and must stay bit-exact."""
    monkeypatch.setenv("TIO_EXACT_LEAN", "2")
    check_case(CASES[name], "cuda")
