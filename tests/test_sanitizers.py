"""SURVEY.md section 5, sanitizer hook: the CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer.
  * tests/cpp/host_sanitize -- the PRODUCT's host side (BMP / PGM / YAML / Gray-table readers fed malformed files, config
    validation, the launch planner over thousands of tile shapes and tuning values, the gather planner for worlds of 1-8,
    slx_create without a device), compiled by g++ from the product's own sources; the kernel launchers are not linked.
  * oracle/asan_driver -- the C restatement on odd shapes, unstructured bytes, exact-size allocations.
Neither needs a GPU.  A sanitizer report aborts the program (non-zero exit) and shows up in stderr."""
import os
import subprocess

from conftest import ROOT


def _run(cmd, cwd=None, timeout=900):
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "abort_on_error=0:detect_leaks=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["LSAN_OPTIONS"] = "suppressions=" + os.path.join(ROOT, "tests", "cpp", "lsan.supp") + ":print_suppressions=0"
    # CPU only, also on a GPU box: with no visible device slx_create returns SLX_ERR_NO_DEVICE before any context exists, so the
    # sanitized process never builds HIP state and the result is the same in the CPU container and on the box
    env["HIP_VISIBLE_DEVICES"] = ""
    env["ROCR_VISIBLE_DEVICES"] = ""
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout, env=env)


def _clean(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    for word in ("AddressSanitizer", "runtime error", "LeakSanitizer", "CHECK failed"):
        assert word not in r.stderr, r.stderr[-6000:]


def test_product_host_side_under_asan_ubsan(tmp_path):
    b = _run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "host_sanitize"])
    assert b.returncode == 0, b.stdout + b.stderr
    r = _run([os.path.join(ROOT, "tests", "cpp", "host_sanitize"), str(tmp_path)])
    _clean(r)
    assert "host_sanitize ok" in r.stdout


def test_oracle_under_asan_ubsan():
    b = _run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    assert b.returncode == 0, b.stdout + b.stderr
    r = _run([os.path.join(ROOT, "oracle", "asan_driver")])
    _clean(r)
    assert "asan_driver ok" in r.stdout
