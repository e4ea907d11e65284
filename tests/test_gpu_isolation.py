"""The `-m gpu` isolation of tests/conftest.py, exercised without a GPU: a test that kills its process (what a device fault does
-- `Memory access fault by GPU`, SIGABRT) must cost one test, not the record of the run (GPUTEST_r05: 0 of 58 reported)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CRASHING = '''
import os, pytest
pytestmark = pytest.mark.gpu
def test_a_ok(): pass
def test_b_dies(): os.abort()
def test_c_ok_after_the_crash(): pass
'''
HEALTHY = '''
import pytest
pytestmark = pytest.mark.gpu
@pytest.mark.parametrize("x", [1, 2])
def test_other_module(x): assert x
def test_skipped(): pytest.skip("no second device")
'''


def test_a_dying_gpu_test_costs_one_test(tmp_path):
    t = tmp_path / "tests"
    t.mkdir()
    shutil.copy(os.path.join(ROOT, "tests", "conftest.py"), t / "conftest.py")
    (t / "test_a_crashing.py").write_text(CRASHING)
    (t / "test_b_healthy.py").write_text(HEALTHY)
    env = {k: v for k, v in os.environ.items() if not k.startswith("RAPID_GPU_")}
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 1, out[-3000:]
    assert "1 failed, 4 passed, 1 skipped" in out, out[-3000:]
    assert "test_b_dies" in out and "DEVICE FAULT or crash" in out and "signal 6" in out, out[-3000:]
    # in-process (the escape hatch) the same run dies with the test
    env["RAPID_GPU_NO_ISOLATION"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode not in (0, 1) and " passed" not in r.stdout
