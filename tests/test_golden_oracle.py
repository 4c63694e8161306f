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
