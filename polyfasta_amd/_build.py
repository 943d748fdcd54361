"""Build libpfa_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libpfa_hip.so")
SOURCES = ["pfa_scan.hip", "pfa_synth.hip", "pfa_fasta.cpp", "pfa_format.cpp"]
HEADERS = ["pfa_codon.h", "pfa_common.h", os.path.join("..", "..", "include", "pfa.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-fconstexpr-steps=500000000", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpfa_hip.so cannot be built (there is no CPU fallback)")


def stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile every HIP/C++ source of the package into polyfasta_amd/libpfa_hip.so."""
    if not force and not stale():
        return LIB_PATH
    cmd = [_hipcc()] + FLAGS + ["-o", LIB_PATH + ".tmp"] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
