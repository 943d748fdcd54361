import ctypes, time, numpy as np, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from polyfasta_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(1)
row = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=10000).astype(np.uint8)
dst = np.zeros(2600, dtype=np.uint8)
n = 200000
f0 = lib.pfa_fasta_pack2
t0 = time.perf_counter()
for _ in range(n):
    f0(_lib.ptr(row), 0, _lib.ptr(dst))
print("empty call: %.2f us" % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(10):
    pass
f = lib.pfa_fasta_pack2; a, b = _lib.ptr(row), _lib.ptr(dst)
t0 = time.perf_counter()
for _ in range(n):
    f(a, 10000, b)
dt = time.perf_counter() - t0
print("pack2_row: %.2f us per 10 kB row incl. ctypes call = %.2f GB/s" % (dt / n * 1e6, n * 10000 / dt / 1e9))
text = (b">s\n" + row.tobytes() + b"\n") * 50
buf = np.frombuffer(text, dtype=np.uint8).copy()
rows = np.zeros(64, dtype=_lib.FASTA_ROW_DTYPE)
g = lib.pfa_fasta_index_compact
t0 = time.perf_counter()
for _ in range(20000):
    g(_lib.ptr(buf), len(buf), _lib.ptr(rows), 64)
dt = time.perf_counter() - t0
print("index_compact: %.2f us per 500 kB file = %.2f GB/s" % (dt / 20000 * 1e6, 20000 * len(buf) / dt / 1e9))
big = np.zeros(500000, dtype=np.uint8); out = np.zeros(500000, dtype=np.uint8)
t0 = time.perf_counter()
for _ in range(20000):
    ctypes.memmove(out.ctypes.data, big.ctypes.data, 125000)
dt = time.perf_counter() - t0
print("memcpy 125 kB: %.2f us" % (dt / 20000 * 1e6))
os.system("lscpu | grep -i 'model name\\|flags' | cut -c1-400")
