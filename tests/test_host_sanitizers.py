"""AddressSanitizer + UBSan over the host code of the library (FASTA packer, batch loader,
CSV formatter) on random inputs.  CPU only: GPU sanitizers are not available on the pool, and
this is the code that parses arbitrary user files."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "polyfasta_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_fuzz"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-pthread", os.path.join(ROOT, "tests", "host_fuzz.cpp"),
           os.path.join(CSRC, "pfa_fasta.cpp"), os.path.join(CSRC, "pfa_format.cpp"), "-o", str(exe)]
    subprocess.check_call(cmd)
    work = tmp_path / "files"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([str(exe), "8000", str(work)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    assert "iterations clean" in out.stdout
