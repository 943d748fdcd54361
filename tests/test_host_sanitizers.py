"""AddressSanitizer + UBSan over the host code of the library (FASTA packer, batch loader,
CSV formatter) on random inputs, and ThreadSanitizer / AddressSanitizer over the streaming
ingest's pipeline (loader threads, pinned ring, uploader, batch assembly) against a threaded
double of the HIP runtime.  CPU only: GPU sanitizers are not available on the pool, and this
is the code that parses arbitrary user files and shares buffers between threads."""
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


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_ingest_pipeline_under_sanitizers(tmp_path, sanitizer):
    """csrc/pfa_ingest.hip's host side built without device code (-DPFA_INGEST_HOST_TEST) against tests/hip_mock.h —
    streams are threads that run their copies later than the call that enqueued them — and driven by
    tests/ingest_tsan.cpp: directories of awkward files (wrapped, CRLF, dirty rows, unequal, not FASTA, unreadable,
    non-ASCII, larger than a slot), 1-16 loaders, 16 KiB-1 MiB slots, feeds in grains, population keys; every matrix
    byte, unit table and status is checked against an independent reading.  (Two bugs of this round were of the kind
    this finds: a pinned buffer handed back twice; the harness reports both when they are put back in.)"""
    exe = tmp_path / "ingest_san"
    cmd = ["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-DPFA_INGEST_HOST_TEST",
           "-I" + os.path.join(ROOT, "tests"), "-pthread", "-x", "c++", os.path.join(CSRC, "pfa_ingest.hip"),
           os.path.join(CSRC, "pfa_fasta.cpp"), os.path.join(ROOT, "tests", "ingest_tsan.cpp"), "-o", str(exe)]
    subprocess.check_call(cmd)
    work = tmp_path / "files"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    env.pop("LD_PRELOAD", None)
    for var in ("PFA_THREADS", "PFA_INGEST_SLOT_BYTES", "PFA_INGEST_SLOTS", "PFA_INGEST_TRANSPORT"):
        env.pop(var, None)
    out = subprocess.run([str(exe), "30", str(work)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "30 rounds clean" in out.stdout
    if sanitizer == "thread":          # text over the link as well (no packing): the other branch of the loaders
        out = subprocess.run([str(exe), "10", str(work)], env=dict(env, PFA_INGEST_TRANSPORT="raw"), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "10 rounds clean" in out.stdout, (out.stdout[-2000:], out.stderr[-6000:])
