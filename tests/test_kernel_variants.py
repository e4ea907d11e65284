"""The prepared compile-time variants of the tally kernel (RAPID_LEAN_V2, RAPID_CAREFUL_HINT, RAPID_EARLY_CERT,
RAPID_FAST_WINDOW, RAPID_DMA_PAIRS; csrc/tally_kernel.h, scripts/ab_lean_v2.sh) must keep producing the oracle's results: the emulated
kernel tests are re-run on the combined builds in a child process (the emulator library is chosen once per process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["all", "all5"])
def test_emulated_kernel_tests_on_variant(variant):
    env = {**os.environ, "RAPID_EMU_VARIANT": variant}
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernel_emulated.py", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "scenarios or churn or unaligned or adversarial"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
