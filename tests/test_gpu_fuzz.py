"""Randomised parity (tests/fuzz_parity.py): random shapes, method mixes, lags, near lags, streamer variants, storages and shard
limits -- the device against the blocked oracle with the layout the library reports, bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_are_bit_exact():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "60", "5"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "0 failures" in out.stdout and out.stdout.count("\nok ") + out.stdout.startswith("ok ") >= 50
