import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from polyfasta_amd import _lib, dist
lib = _lib.load()
print("domains (whole mask):", lib.pfa_host_l3_domains(None, None, 0), "cpus", len(os.sched_getaffinity(0)))
print("bound to", len(dist.bind_to_gpu_node(0, 1) or []))
d = np.full(256, -1, dtype=np.int32)
n = lib.pfa_host_l3_domains(None, _lib.ptr(d), 256)
print("domains after binding:", n, d[:20].tolist(), d[60:70].tolist())
print(open("/sys/devices/system/cpu/cpu0/cache/index3/id").read().strip(), open("/sys/devices/system/cpu/cpu64/cache/index3/id").read().strip())
