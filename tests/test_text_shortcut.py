"""The double-arithmetic shortcut of the device formatter (csrc/slx_text.hip) against the exact integer arithmetic it stands in front of, on
the host (tests/cpp/text_shortcut.cpp: the same IEEE operations): random magnitudes, the neighbourhoods of the ties of the sixth digit,
of the powers of ten and of the integers.  The GPU test test_point_cloud_text_formatted_on_the_device compares the device's bytes themselves."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shortcut_equals_the_exact_integers(tmp_path):
    src = os.path.join(ROOT, "tests", "cpp", "text_shortcut.cpp")
    exe = str(tmp_path / "text_shortcut")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", src, "-o", exe])
    out = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-500:]
    m = re.search(r"text_shortcut: (\d+) cases, (\d+) by the shortcut, 0 differences", out.stdout)
    assert m and int(m.group(1)) > 90_000_000 and int(m.group(2)) > 3_000_000, out.stdout


def test_the_checked_arithmetic_is_the_kernel_s():
    """The host program restates the kernel's shortcut: the lines that decide must be the same text in both files."""
    k = open(os.path.join(ROOT, "structured-light-calculation_amd", "csrc", "slx_text.hip")).read()
    h = open(os.path.join(ROOT, "tests", "cpp", "text_shortcut.cpp")).read()
    for line in ("const double P = ((p & 1) ? 10.0 : 1.0) * ((p & 2) ? 100.0 : 1.0) * (((p & 4) ? 1e4 : 1.0) * ((p & 8) ? 1e8 : 1.0));",
                 "const double t = a * P;",
                 "> 0x1p-30 &&", "- 1e5) > 1e-6 &&", "- 1e6) > 1e-6) {",
                 "if (f < 1e5) { X--; continue; }", "if (f >= 1e6) { X++; continue; }",
                 "unsigned q32 = (unsigned)f + (fr > 0.5 ? 1u : 0u);", "if (q32 == 1000000u) { q32 = 100000u; X++; }"):
        assert line in k and line in h, line
