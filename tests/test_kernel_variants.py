"""The tally kernel's window size is a compile-time constant (RAPID_QUARTERS x 64 records, csrc/tally_kernel.h); the
small populations of the emulated tests hold only a few hundred records per receiver, so the emulated kernel tests are
re-run on builds with 64-, 128- and 192-record windows (every stream then spans many cold / fast / slow windows and carries)
and on the sanitizer build, in a child process (the emulator library is chosen once per process).  (The per-slot tables --
masks of the rings on which a hot observer watches a slot, slot -> node -- are read from memory by the instantiations with the
packed detector state, the code of rounds with more than 4,096 hot subjects: tests/test_kernel_emulated.py runs those at the
emulator's population sizes in every build.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["q1", "q2", "q3", "ubsan"])
def test_emulated_kernel_tests_on_variant(variant):
    env = {**os.environ, "RAPID_EMU_VARIANT": variant}
    # (the window-size variants change the tally kernel only: the index / vote / view kernel tests of that file run once, in
    # the default build -- except under the sanitizer, where everything runs)
    only = [] if variant == "ubsan" else ["-k", "not (round_index or vote_ or votes_ or view_kernels or view_change or identifier or joiner)"]
    # (the child run spreads over a few workers where pytest-xdist is there: the variants are most of the CPU suite's wall time)
    try:
        import xdist  # noqa: F401
        workers = ["-n", str(max(1, min(4, (os.cpu_count() or 2) // 2)))]
    except ImportError:
        workers = []
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernel_emulated.py", "-x", "-q", "-p", "no:cacheprovider"] + workers + only,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
