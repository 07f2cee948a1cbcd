"""ASan + UBSan over the device per-read code compiled for the host (tests/host_emul): GPU
sanitizers are not available on the pool, so memory / UB checking of the kernel logic happens
on this CPU build."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_device_code_under_asan_ubsan():
    lib = _libasan()
    if lib is None:
        pytest.skip("libasan not found")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host_emul"), "build/libdcrx_emul_asan.so"],
                          stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=lib, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "ASAN_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr, p.stderr[-4000:]
