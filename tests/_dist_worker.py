"""Worker of tests/test_dist_cpu.py: two gloo ranks exercise the multi-GPU host path
(size-aware sharding, the fixed-size record all-gather, the pickled metadata, the
file-ordered merge on rank 0) with synthetic records — no GPU involved."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from polyfasta_amd import _lib, cli, dist  # noqa: E402
from tests._gloo_comm import GlooComm  # noqa: E402


def fake_events(paths, indices, cds):
    """What process_files would return for these files: file i has i % 3 + 1 units
    (one per population), every third file also a comment line."""
    evs = []
    for i in indices:
        name = paths[i].split("/")[-1]
        if i % 3 == 0:
            evs.append((i, "comment", f"# CDS sequence length is not a multiple of 3: {name}"))
        if i % 5 == 4:
            evs.append((i, "stdout", f"# file {paths[i]} is not FASTA!"))
            continue
        for k in range(i % 3 + 1):
            rec = np.zeros(1, dtype=_lib.RESULT_DTYPE)[0]
            rec["seg_sites"], rec["H"], rec["seg_S"], rec["nstops"] = 10 * i + k, 1000 * i + k, i, k
            syn = np.arange(_lib.PFA_SYN_BINS, dtype=np.uint32) * (i + 1) + k if cds else None
            sfs = [i, k, 7] if i % 2 else None
            evs.append((i, "unit", name, f"pop{k}", 20 + k, 3000 + i, rec, syn, sfs))
    return evs


def main():
    out_path = sys.argv[1]
    n_files = int(sys.argv[2])
    g = GlooComm()
    paths = sorted(f"dir/file{i}.fa" for i in range(n_files))
    sizes = [1000 + 37 * (i % 11) for i in range(n_files)]
    mine = dist.shard_by_size(sizes, g.world)[g.rank]
    ok = True
    for cds in (False, True):
        merged = cli.gather_events(g, fake_events(paths, mine, cds))
        if g.rank == 0:
            want = fake_events(paths, range(n_files), cds)
            ok = ok and len(merged) == len(want)
            for a, b in zip(merged, want):
                ok = ok and a[:6] == b[:6]
                if a[1] == "unit":
                    ok = ok and tuple(a[6]) == tuple(b[6].tolist()) and a[8] == b[8]
                    ok = ok and ((a[7] is None and b[7] is None) or (a[7] == b[7]).all())
    blobs = g.all_gather_bytes(b"x" * (5 + 100 * g.rank))
    ok = ok and [len(b) for b in blobs] == [5 + 100 * r for r in range(g.world)]
    tmax = g.max_float(1.0 + g.rank)
    g.barrier()
    if g.rank == 0:
        json.dump({"ok": bool(ok), "n": len(merged), "world": g.world, "tmax": tmax}, open(out_path, "w"))
    g.close()


if __name__ == "__main__":
    main()
