"""examples/ run end to end on the device (no oracle involved: the product's prover and the product's verifier)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_simple_example_proves_and_verifies():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "simple_example.py")
    spec = importlib.util.spec_from_file_location("simple_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main([]) is True
    assert mod.main(["--k", "6"]) is True
