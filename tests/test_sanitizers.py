"""Sanitizer runs (race / sanitizer row), CPU only.

The reference guards its shared state with `mp.RLock` (denet/multi/shared.py:28-48) and an `atomic<bool>`
(denet/layer/denet_sparse.cc:491); this build keeps no shared mutable state across threads on the hot path, but its host-native
functions (MT19937 emulation, RoI list editing, detection targets, RoI clustering, soft-NMS, Pillow coefficient tables:
csrc/runtime.hip, samples.hip, detect.hip, image.hip) do raw pointer arithmetic on caller buffers. They are compiled with
AddressSanitizer + UndefinedBehaviorSanitizer on the host side and driven by tests/native/host_sanitize.cc with exact-size heap
buffers (trim / no trim / empty ground truth / 39 boxes / cluster thresholds / capacity errors); the oracle's C++ checker gets the
same treatment (`make -C oracle asan`)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _run(cmd, cwd, timeout):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "%s\n%s\n%s" % (" ".join(cmd), r.stdout[-3000:], r.stderr[-6000:])
    return r.stdout + r.stderr


def test_oracle_checker_under_asan_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], ROOT, 600)
    assert "oracle_asan: ok" in out and "runtime error" not in out and "AddressSanitizer" not in out


def test_host_native_functions_under_asan_ubsan():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc: the host-native functions live in .hip files")
    out = _run(["make", "-s", "-C", os.path.join(ROOT, "tests", "native"), "run", "HIPCC=" + HIPCC], ROOT, 1200)
    assert "host_sanitize: ok" in out and "runtime error" not in out and "AddressSanitizer" not in out
