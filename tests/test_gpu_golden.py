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
