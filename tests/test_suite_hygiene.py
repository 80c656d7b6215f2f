"""The suite's own switches (CPU)."""
import types

import pytest


@pytest.mark.parametrize("expr,expected", [("gpu", False), ("gpu and sweep", True), ("gpu and not sweep", False), ("not sweep and gpu", False),
                                           ("sweep", True), ("gpu and not (sweep)", False), ("", False), ("gpu and not sweep or sweep", True)])
def test_sweep_mode_reads_the_marker_expression(expr, expected, monkeypatch):
    """`-m "gpu and not sweep"` must NOT run the exhaustive grids"""
    from conftest import sweep_mode
    monkeypatch.delenv("SNN_TEST_SWEEP", raising=False)
    assert sweep_mode(types.SimpleNamespace(getoption=lambda k: expr)) is expected
    monkeypatch.setenv("SNN_TEST_SWEEP", "1")
    assert sweep_mode(types.SimpleNamespace(getoption=lambda k: expr)) is True
