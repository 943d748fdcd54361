"""Build libpfa_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to an object on its own (in parallel: the kernel translation
units take most of a minute each) and the objects are linked into one shared library."""
import os

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
    import shutil
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpfa_hip.so cannot be built (there is no CPU fallback)")


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _unit_hash(src, flags=()):
    """what an object file of `src` depends on: the source, every header, the compile line (contents, not times)"""
    import hashlib
    h = hashlib.sha256(" ".join(CFLAGS + list(flags)).encode())
    for f in [src] + sorted(HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(b"\0" + fh.read())
    return h.hexdigest()


def _obj_stale(o, want):
    try:
        with open(o + ".id") as fh:
            return not os.path.exists(o) or fh.read().strip() != want
    except OSError:
        return True


def source_hash():
    """16 hex digits over the contents of every source and header the library is compiled from (names and bytes, fixed
    order): what pfa_build_id() of a library built from this tree returns.  None when the tree holds no sources (an
    installed copy of the built library alone)."""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in sorted(SOURCES + HEADERS):
            with open(os.path.join(CSRC, f), "rb") as fh:
                data = fh.read()
            h.update(os.path.basename(f).encode() + b"\0" + str(len(data)).encode() + b"\0" + data)
    except OSError:
        return None
    return h.hexdigest()[:16]


def built_id(path=None):
    """the build id inside a library file, read without loading it (the bytes after the marker), or None"""
    try:
        with open(path or LIB_PATH, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    k = blob.find(b"PFA_BUILD_ID=")
    return blob[k + 13:k + 29].decode("ascii", "replace") if k >= 0 else None


def stale():
    """the library is missing or was built from other sources than the tree's (contents, not modification times: a checkout
    or a copy that refreshes the times must not pair new sources with an old library)"""
    if not os.path.exists(LIB_PATH):
        return True
    want = source_hash()
    return want is not None and built_id() != want


def build(force=False, verbose=False, extra_flags=(), lib_path=None):
    """Compile every HIP/C++ source of the package into polyfasta_amd/libpfa_hip.so.
    `extra_flags` (e.g. -D switches of a timing experiment) force a full rebuild into `lib_path`."""
    out = lib_path or LIB_PATH
    if not force and not extra_flags and out == LIB_PATH and not stale():
        return out
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    import hashlib
    tag = "" if not extra_flags else "." + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]
    tmp = f".{os.getpid()}.tmp"      # two builds at once (two ranks, two tools) never share a temporary

    def compile_one(src):
        o = _obj(src) if not tag else _obj(src)[:-2] + tag + ".o"
        want = _unit_hash(src, extra_flags)
        if not force and not _obj_stale(o, want):
            return o
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [hipcc] + CFLAGS + list(extra_flags) + ["-c"] + lang + ["-o", o + tmp, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(o + tmp, o)
        with open(o + ".id" + tmp, "w") as fh:
            fh.write(want)
        os.replace(o + ".id" + tmp, o + ".id")
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # the sources' hash, compiled in (pfa_build_id): one generated translation unit, rebuilt with every link
    idsrc = os.path.join(OBJ_DIR, f"pfa_build_id{tag}.cpp")
    with open(idsrc + tmp, "w") as fh:
        fh.write('extern "C" const char pfa_build_id_string[] = "PFA_BUILD_ID=%s";\n' % source_hash())
    os.replace(idsrc + tmp, idsrc)
    ido = idsrc[:-4] + ".o"
    subprocess.check_call([hipcc, "-O1", "-fPIC", "-c", "-o", ido + tmp, idsrc])
    os.replace(ido + tmp, ido)
    objs.append(ido)
    cmd = [hipcc] + LDFLAGS + ["-o", out + tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + tmp, out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
