"""AddressSanitizer + UndefinedBehaviorSanitizer build of the library's host-side C (csrc/host_util.cpp) with a driver that
checks radnet_host_choice_round against a plain restatement of NumPy's choice round on exactly-sized buffers
(tests/native/host_util_sanitize.cpp).  CPU only: GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_util_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_util_sanitize")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "rock-art-radnet_amd", "csrc", "host_util.cpp"), os.path.join(ROOT, "tests", "native", "host_util_sanitize.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
