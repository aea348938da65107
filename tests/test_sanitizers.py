"""CPU suite: the host-side code that handles caller data (mesh loaders, SAH builder, 4-wide collapse, scene generators) and
the oracle, built with AddressSanitizer + UBSan and run over well-formed, truncated and garbage mesh files (`make asan`)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_and_ubsan():
    res = subprocess.run(["make", "asan"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "sanitizer run clean" in res.stdout
