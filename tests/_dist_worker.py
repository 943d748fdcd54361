"""Worker of tests/test_dist_cpu.py: two gloo ranks exercise the multi-GPU
host path (round-robin sharding, the fixed-size record all-gather, the
file-ordered merge on rank 0) with synthetic records — no GPU involved."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from polyfasta_amd import _lib, cli, dist  # noqa: E402


def fake_events(paths, indices):
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
            syn = np.arange(_lib.PFA_SYN_BINS, dtype=np.uint32) * (i + 1) + k
            evs.append((i, "unit", name, f"pop{k}", 20 + k, 3000 + i, rec, syn))
    return evs


def main():
    out_path = sys.argv[1]
    n_files = int(sys.argv[2])
    g = dist.Group(backend="gloo")
    paths = sorted(f"dir/file{i}.fa" for i in range(n_files))
    mine = dist.shard_indices(n_files, g.rank, g.world)
    merged = cli.gather_events(g, fake_events(paths, mine))
    tmax = g.max_float(1.0 + g.rank)
    g.barrier()
    if g.rank == 0:
        want = fake_events(paths, range(n_files))
        ok = len(merged) == len(want)
        for a, b in zip(merged, want):
            ok = ok and a[:6] == b[:6]
            if a[1] == "unit":
                ok = ok and tuple(a[6]) == tuple(b[6].tolist()) and (a[7] == b[7]).all()
        json.dump({"ok": bool(ok), "n": len(merged), "world": g.world, "tmax": tmax}, open(out_path, "w"))
    g.close()


if __name__ == "__main__":
    main()
