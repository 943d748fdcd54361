"""Worker of tests/test_dist_cpu.py: two gloo ranks exercise the multi-GPU host path
(shards of a file list, row blocks gathered to rank 0 as one byte string per rank, the
file-ordered merge there) with synthetic records — no GPU involved."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from polyfasta_amd import _lib, cli, dist  # noqa: E402
from tests._gloo_comm import GlooComm  # noqa: E402

KEYS = ["pop0", "pop1", "pop2"]


def fake_block(paths, indices, cds, sfs):
    """What run_files would keep for these files: file i has i % 3 + 1 units (one per
    population), every third file also a comment line, some files are not FASTA."""
    meta, recs, syns, sf, specials = [], [], [], [], []
    for i in indices:
        name = paths[i].split("/")[-1]
        if i % 3 == 0:
            specials.append((i, 0, "comment", f"# CDS sequence length is not a multiple of 3: {name}"))
        if i % 5 == 4:
            specials.append((i, 0, "stdout", f"# file {paths[i]} is not FASTA!"))
            continue
        for k in range(i % 3 + 1):
            recs.append((1000 * i + k, 0, 0, 10 * i + k, 0, i, 0, k, 0, 0, 0))
            syns.append(np.arange(_lib.PFA_SYN_BINS, dtype=np.uint32) * (i + 1) + k)
            sf.append([i, k, 7] if i % 2 else None)
            meta.append((i, k, 20 + k, 3000 + i))
    return cli.RowBlock(np.array(meta, dtype=cli.META_DTYPE).reshape(-1), np.array(recs, dtype=_lib.RESULT_DTYPE).reshape(-1),
                        (np.stack(syns) if syns else np.zeros((0, _lib.PFA_SYN_BINS), dtype=np.uint32)) if cds else None,
                        sf if sfs else None, specials)


def main():
    out_path = sys.argv[1]
    n_files = int(sys.argv[2])
    g = GlooComm()
    paths = sorted(f"dir/file{i}.fa" for i in range(n_files))
    sizes = [1000 + 37 * (i % 11) for i in range(n_files)]
    mine = dist.shard_by_size(sizes, g.world)[g.rank]
    ok, n = True, 0
    for cds in (False, True):
        for sfs in (False, True):
            # two blocks per rank, the later files first: rank 0 must order them
            half = len(mine) // 2
            local = cli.RowBlock.concat([fake_block(paths, mine[half:], cds, sfs), fake_block(paths, mine[:half], cds, sfs)], cds)
            parts = g.gather_bytes(local.to_bytes(), dst=0)
            if g.rank != 0:
                ok = ok and parts is None
                continue
            merged = cli.block_events(cli.RowBlock.concat([cli.RowBlock.from_bytes(p) for p in parts], cds).sorted(), paths, KEYS)
            want = cli.block_events(fake_block(paths, range(n_files), cds, sfs), paths, KEYS)
            n = len(merged)
            ok = ok and len(merged) == len(want)
            for a, b in zip(merged, want):
                ok = ok and a[:6] == b[:6]
                if a[1] == "unit":
                    ok = ok and tuple(a[6]) == tuple(b[6]) and a[8] == b[8]
                    ok = ok and ((a[7] is None and b[7] is None) or (a[7] == b[7]).all())
    blobs = g.all_gather_bytes(b"x" * (5 + 100 * g.rank))
    ok = ok and [len(b) for b in blobs] == [5 + 100 * r for r in range(g.world)]
    tmax = g.max_float(1.0 + g.rank)
    g.barrier()
    if g.rank == 0:
        json.dump({"ok": bool(ok), "n": n, "world": g.world, "tmax": tmax}, open(out_path, "w"))
    g.close()


if __name__ == "__main__":
    main()
