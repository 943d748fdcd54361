"""Build libpfa_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to an object on its own (in parallel: the kernel translation
units take most of a minute each) and the objects are linked into one shared library."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.environ.get("PFA_LIB") or os.path.join(_HERE, "libpfa_hip.so")
SOURCES = ["pfa_scan_classic.hip", "pfa_scan_tall.hip", "pfa_plan.hip", "pfa_synth.hip", "pfa_ingest.hip", "pfa_comm.cpp",
           "pfa_fasta.cpp", "pfa_format.cpp"]
HEADERS = ["pfa_codon.h", "pfa_common.h", "pfa_host.h", "pfa_kernels.h", "pfa_launch.h", "pfa_layout.h",
           os.path.join("..", "..", "include", "pfa.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fconstexpr-steps=500000000", "-Wall",
          "-Wno-unused-function"]
FLAGS = CFLAGS + ["-shared"]      # (tools/kernel_resources.py derives its compile line from this)
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl", "-lpthread"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpfa_hip.so cannot be built (there is no CPU fallback)")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _obj_stale(src, t_hdr):
    o = _obj(src)
    return not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(os.path.join(CSRC, src)), t_hdr)


def stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra_flags=(), lib_path=None):
    """Compile every HIP/C++ source of the package into polyfasta_amd/libpfa_hip.so.
    `extra_flags` (e.g. -D switches of a timing experiment) force a full rebuild into `lib_path`."""
    out = lib_path or LIB_PATH
    if not force and not extra_flags and out == LIB_PATH and not stale():
        return out
    os.makedirs(OBJ_DIR, exist_ok=True)
    t_hdr = _newest_header()
    hipcc = _hipcc()
    import hashlib
    tag = "" if not extra_flags else "." + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]
    tmp = f".{os.getpid()}.tmp"      # two builds at once (two ranks, two tools) never share a temporary

    def compile_one(src):
        o = _obj(src) if not tag else _obj(src)[:-2] + tag + ".o"
        if not (force or extra_flags) and not _obj_stale(src, t_hdr):
            return o
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [hipcc] + CFLAGS + list(extra_flags) + ["-c"] + lang + ["-o", o + tmp, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(o + tmp, o)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc] + LDFLAGS + ["-o", out + tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + tmp, out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
