"""CPU: the HIP-free host logic of the engine (histogram compaction from the device tables, the host histogram pass kept as
its checker, the task list of the histogram kernel) built with AddressSanitizer + UndefinedBehaviorSanitizer and run
(sanitizers belong on the CPU build; the GPU pool has none)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_logic_sanitize")
    src = os.path.join(ROOT, "tests", "host_logic_sanitize.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", src, "-o", exe, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "host logic ok" in r.stdout
