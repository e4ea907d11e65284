"""The JNI binding of INTEGRATION.md as a real file (jni/rapid_mi355x_jni.c): type-checked against
include/rapid_mi355x.h with a stub jni.h (no JDK in this image), and its host-only natives -- NativeFastPaxos and the
request decoder -- executed under a fake JNIEnv: five nodes run a fast round without quorum and a classic recovery round
exchanging only the serialized RapidRequests the shim hands out.  Host only."""
import os
import shutil
import subprocess

import pytest

from rapid_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, "tests", "jni_stub"), "-I" + os.path.join(ROOT, "include")]

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")


def test_shim_type_checks_against_the_header():
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", *INC,
                           os.path.join(ROOT, "jni", "rapid_mi355x_jni.c")])


def test_consensus_natives_run_under_a_fake_jvm(tmp_path):
    lib_dir = os.path.dirname(N.LIB_PATH)
    N.lib()  # the product library must be built
    exe = str(tmp_path / "fake_jvm")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-Wall", "-Wextra", *INC, os.path.join(ROOT, "jni", "rapid_mi355x_jni.c"),
                           os.path.join(ROOT, "tests", "jni_stub", "fake_jvm_main.c"), "-L" + lib_dir, "-l:" + os.path.basename(N.LIB_PATH),
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "through the JNI shim" in r.stdout
