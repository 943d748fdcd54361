"""Randomised GPU-vs-oracle parity over shapes, alphabets, strides and row
subsets (hash-seeded, reproducible).  Integers must match exactly."""
import numpy as np
import pytest

from oracle import c_oracle
from polyfasta_amd import scan, synth
from tests import _cases as C

pytestmark = pytest.mark.gpu

ALPHABETS = [b"ACGT", b"ACGT-", b"ACGTN-", b"ACGTRYKMSWBDHVN-?", b"AC", b"-N", b"ACGTU*.@EFPQVW"]


def _random_matrix(rng, n, L, alphabet, p_mut):
    """consensus row + per-cell substitutions at rate p_mut: many monomorphic
    columns, some multi-allelic, some all-odd-symbol columns"""
    a = np.frombuffer(alphabet, dtype=np.uint8)
    cons = a[rng.integers(0, len(a), size=L)]
    m = np.tile(cons, (n, 1))
    hit = rng.random((n, L)) < p_mut
    m[hit] = a[rng.integers(0, len(a), size=int(hit.sum()))]
    return m


def _compare(out, i, o, cds):
    r = out.results[i]
    assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"])
    assert out.sfs(i).tolist() == o["sfs"].tolist()
    assert out.var_positions(i).tolist() == np.flatnonzero(o["h_site"]).tolist()
    if cds:
        assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
               (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"])
        assert out.syn_positions(i).tolist() == np.flatnonzero(o["labels"] == 1).tolist()
        assert out.non_positions(i).tolist() == np.flatnonzero(o["labels"] == 2).tolist()
        from polyfasta_amd import stats
        assert abs(stats.syn_sites_from_bins(out.syn_num[i]) - o["count_syn"]) <= 1e-9 * max(1.0, o["count_syn"])


@pytest.mark.parametrize("seed", range(6))
def test_random_batches(gpu, seed):
    rng = np.random.default_rng(1234 + seed)
    b = scan.Batch(stride_align=[16, 32, 128, 256, 16, 64][seed])
    units = []
    for _ in range(60):
        n = int(rng.choice([1, 2, 3, 4, 5, 8, 17, 33, 50, 64, 65, 100, 129, 255, 300]))
        L = int(rng.choice([1, 2, 3, 4, 15, 16, 17, 47, 48, 49, 63, 64, 65, 255, 256, 257, 511, 512, 513, 767, 768, 769, 770, 1000, 1537, 3001]))
        alpha = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))]
        m = _random_matrix(rng, n, L, alpha, float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3])))
        mid = b.add_matrix(m)
        for cds in (False, True):
            b.add_unit(mid, cds=cds)
            units.append((m, None, cds))
        if n >= 3:
            rows = np.sort(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False)).astype(np.uint32)
            if rng.random() < 0.3:
                rows = rows[::-1].copy()
            cds = bool(rng.random() < 0.5)
            b.add_unit(mid, rows=rows, cds=cds)
            units.append((m, rows, cds))
    out = scan.scan_batch(b, want_pos=True, want_sn=True)
    for i, (m, rows, cds) in enumerate(units):
        _compare(out, i, c_oracle.scan(m, rows=rows, cds=cds, want_sites=True), cds)


@pytest.mark.parametrize("seed,cds", [(0, False), (1, True), (2, True), (3, True)])
def test_random_batches_lean_kernels(gpu, seed, cds):
    """Plans without row subsets and without bitmaps run the lean instantiations (the batch
    path of the CLI and the bench): same integers as the oracle."""
    rng = np.random.default_rng(777 + seed)
    b = scan.Batch(stride_align=[128, 16, 128, 64][seed])
    units = []
    for _ in range(80):
        n = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 50, 64]))      # coding or not, at most 64 rows: scan_kernel<4, ., ., true>
        L = int(rng.choice([1, 2, 3, 15, 16, 17, 255, 256, 257, 258, 511, 512, 513, 767, 768, 769, 770, 1000, 1537, 3001, 9000]))
        alpha = ALPHABETS[int(rng.integers(0, 4))]
        m = _random_matrix(rng, n, L, alpha, float(rng.choice([0.0, 0.001, 0.01, 0.02, 0.05, 0.3])))
        b.add_unit(b.add_matrix(m), cds=cds)
        units.append(m)
    out = scan.scan_batch(b)
    from polyfasta_amd import stats
    for i, m in enumerate(units):
        o = c_oracle.scan(m, cds=cds)
        r = out.results[i]
        assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"]), (i, m.shape)
        assert out.sfs(i).tolist() == o["sfs"].tolist()
        if cds:
            assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
                   (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"]), (i, m.shape)
            assert abs(stats.syn_sites_from_bins(out.syn_num[i]) - o["count_syn"]) <= 1e-9 * max(1.0, o["count_syn"])


def test_random_many_rows(gpu):
    """coding plans with many-row units run the row-chunked kernel for every unit (coding pass fed from
    the slot tables, joint codon columns read back from global memory)"""
    rng = np.random.default_rng(99)
    b = scan.Batch()
    units = []
    for n, L in ((700, 130), (50, 1000), (2048, 65), (3, 64), (1025, 385)):
        m = _random_matrix(rng, n, L, b"ACGTN-", 0.01)
        mid = b.add_matrix(m)
        for cds in (False, True):
            b.add_unit(mid, cds=cds)
            units.append((m, None, cds))
    out = scan.scan_batch(b, want_pos=True, want_sn=True)
    for i, (m, rows, cds) in enumerate(units):
        _compare(out, i, c_oracle.scan(m, rows=rows, cds=cds, want_sites=True), cds)


def _tall_batch(rng, shapes, p_muts, subsets=True):
    b = scan.Batch(stride_align=int(rng.choice([16, 128])))
    units = []
    for n, L in shapes:
        alpha = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))]
        m = _random_matrix(rng, n, L, alpha, float(rng.choice(p_muts)))
        mid = b.add_matrix(m)
        b.add_unit(mid)
        units.append((m, None, False))
        if subsets and n >= 3:
            rows = rng.permutation(n)[: int(rng.integers(1, n + 1))].astype(np.uint32)
            b.add_unit(mid, rows=rows)
            units.append((m, rows, False))
    return b, units


@pytest.mark.parametrize("seed", range(4))
def test_random_tall_units(gpu, seed):
    """non-coding plans whose largest unit has more than 64 rows run the row-chunked kernel for
    every unit: chunk boundaries, sites that turn variable in a late chunk, odd symbols and
    slot overflow (column path), row subsets in any order, rows beyond the old 2 048 limit"""
    rng = np.random.default_rng(4321 + seed)
    big = [(3000, 300), (5000, 70)][seed % 2]
    shapes = [(int(rng.choice([65, 66, 96, 97, 100, 127, 128, 129, 191, 200, 255, 256, 300, 700])),
               int(rng.choice([1, 3, 16, 255, 256, 257, 513, 1000, 3001]))) for _ in range(24)]
    shapes += [(1, 40), (2, 300), (50, 1000), big]
    b, units = _tall_batch(rng, shapes, [0.0, 0.0005, 0.01, 0.05, 0.5])
    bases, table = b.pack()
    plan = scan.Plan(table, want_pos=True)
    assert plan.rows_per_chunk > 0          # the row-chunked kernel was selected
    plan.close()
    out = scan.scan_batch(b, want_pos=True)
    for i, (m, rows, cds) in enumerate(units):
        _compare(out, i, c_oracle.scan(m, rows=rows, want_sites=True), False)


@pytest.mark.parametrize("rows_per_chunk,slots", [(8, 16), (12, 256), (64, 96), (None, 16), (None, 64)])
def test_tall_kernel_forced_on_small_units(gpu, monkeypatch, rows_per_chunk, slots):
    """the same kernel forced onto units of any size (timing-experiment overrides), tiny chunks
    and a tiny slot table: many chunks per tile and frequent slot overflow; rows_per_chunk None: the
    plan's own 48-row chunks (the straight-line code of whole chunks) and a shorter last one"""
    monkeypatch.setenv("PFA_TALL_MIN", "0")
    if rows_per_chunk is not None:
        monkeypatch.setenv("PFA_TALL_ROWS", str(rows_per_chunk))
    monkeypatch.setenv("PFA_TALL_SLOTS", str(slots))
    rng = np.random.default_rng(7 + (rows_per_chunk or slots))
    shapes = [(int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 31, 33, 50, 64, 65, 100, 130])),
               int(rng.choice([1, 5, 64, 255, 256, 257, 770, 1537]))) for _ in range(40)]
    b, units = _tall_batch(rng, shapes, [0.0, 0.002, 0.02, 0.3])
    plan = scan.Plan(b.pack()[1])
    assert 0 < plan.rows_per_chunk <= max(rows_per_chunk or 48, 8)
    plan.close()
    out = scan.scan_batch(b, want_pos=True)
    for i, (m, rows, cds) in enumerate(units):
        _compare(out, i, c_oracle.scan(m, rows=rows, want_sites=True), False)


def test_row_limit(gpu):
    """no limit below the 32-bit counters' (the reference has none, PolyFastA.py:284-315): coding units
    above 2 048 rows run the wide instantiation of the row-chunked kernel; PFA_MAX_ROWS is refused"""
    from polyfasta_amd import _lib
    m = np.full((2049, 33), 65, dtype=np.uint8)
    m[7, 3] = 67
    for cds in (False, True):
        r = scan.scan_matrix(m, cds=cds).results[0]
        assert (int(r["seg_sites"]), int(r["H"])) == (1, 2049 * 2049 - 2048 * 2048 - 1)
    o = c_oracle.scan(m, cds=True)
    assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["missing_cols"])) == \
           (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["missing_cols"])
    d = np.zeros(1, dtype=_lib.DESC_DTYPE)
    d["n_rows"], d["n_sites"], d["row_stride"], d["mat_rows"], d["row_idx_off"] = _lib.PFA_MAX_ROWS + 1, 16, 16, _lib.PFA_MAX_ROWS + 1, _lib.PFA_NO_ROW_IDX
    with pytest.raises(_lib.PfaError) as e:
        scan.Plan(scan.finalize_layout(d, None, 16 * (_lib.PFA_MAX_ROWS + 1)))
    assert e.value.code == -5


def _many_row_matrix(rng, n, L, alphabet):
    """every site: 1-5 alleles at random frequencies (so that second-largest counts land far above the 256
    SFS bins the row-chunked kernel keeps in LDS), a few sites with rare third / fourth alleles"""
    m = np.empty((n, L), dtype=np.uint8)
    syms = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    for s in range(L):
        k = int(rng.integers(1, 6))
        al = rng.choice(syms, size=k)            # (repeats merge: fewer alleles)
        w = rng.random(k) ** 3 + 1e-4
        m[:, s] = rng.choice(al, size=n, p=w / w.sum())
    return m


@pytest.mark.parametrize("cds", [False, True])
def test_units_beyond_65536_rows_match_the_oracle(gpu, cds):
    """What the ABI advertises (PFA_MAX_ROWS = 2^24; the reference has no limit, PolyFastA.py:252-315) against
    the oracle where N^2 and c^2 leave 32 bits: a 70 000-row unit with 3+ alleles per site, gaps, N and IUPAC
    bytes — 64-bit H on the device, SFS bins above the 256 kept in LDS, the 56-bit entries of the coding pass."""
    rng = np.random.default_rng(77 + cds)
    n, L = 70000, 66 if cds else 64
    m = _many_row_matrix(rng, n, L, "ACGT" * 3 + "N-RY")
    m[:, 5] = ord("A")                       # a monomorphic site, a two-allele site with one rare row
    m[:, 6] = ord("C")
    m[123, 6] = ord("T")
    out = scan.scan_matrix(m, cds=cds)
    o = c_oracle.scan(m, cds=cds)
    r = out.results[0]
    assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"])
    assert int(r["H"]) > 2 ** 32 and out.sfs(0).tolist() == o["sfs"].tolist()
    assert int(np.flatnonzero(out.sfs(0)).max()) > 256
    if cds:
        assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
               (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"])
    # a row subset of it (--pops on a huge alignment)
    rows = np.arange(0, n, 1, dtype=np.uint32)[rng.random(n) < 0.95]
    out = scan.scan_matrix(m, rows=rows, cds=cds)
    o = c_oracle.scan(np.ascontiguousarray(m[rows]), cds=cds)
    r = out.results[0]
    assert (int(r["seg_sites"]), int(r["H"])) == (o["S"], o["H"]) and out.sfs(0).tolist() == o["sfs"].tolist()


def test_unit_at_the_row_limit(gpu):
    """PFA_MAX_ROWS = 2^24 rows x 16 sites, coding and not, and a 97 % row subset through an index list, against the C oracle"""
    from polyfasta_amd import _lib
    rng = np.random.default_rng(9)
    n = _lib.PFA_MAX_ROWS
    assert n == 1 << 24
    m = _many_row_matrix(rng, n, 16, "ACGTN")
    o = c_oracle.scan(m)
    r = scan.scan_matrix(m).results[0]
    assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"])
    assert int(r["H"]) > 2 ** 48
    rows = np.flatnonzero(rng.random(n) < 0.97).astype(np.uint32)
    o = c_oracle.scan(m, rows=rows)
    out = scan.scan_matrix(m, rows=rows)
    r = out.results[0]
    assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"]) and out.sfs(0).tolist() == o["sfs"].tolist()
    m = _many_row_matrix(rng, n, 15, "ACGT")
    o = c_oracle.scan(m, cds=True)
    out = scan.scan_matrix(m, cds=True)
    r = out.results[0]
    assert (int(r["seg_sites"]), int(r["H"]), int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"])) == \
           (o["S"], o["H"], o["seg_S"], o["seg_N"], o["H_S"], o["H_N"])
    assert out.sfs(0).tolist() == o["sfs"].tolist()


@pytest.mark.parametrize("rows_per_chunk,slots,seed", [(8, 16, 0), (12, 64, 1), (36, 64, 2), (8, 256, 3), (64, 32, 4),
                                                       (None, 64, 5), (None, 16, 6), (None, 64, 7)])
def test_tall_kernel_coding_forced_on_small_units(gpu, monkeypatch, rows_per_chunk, slots, seed):
    """The coding variant of the row-chunked kernel forced onto units of any size, with tiny chunks
    (columns turn joint in early, middle and late chunks: codon slots seeded from count slots in
    every mode), tiny count-slot tables (positions without usable counts: read-back), mutation
    rates from 0 to 30 % (more joint columns per tile than codon slots: read-back), odd alphabets
    (unclean codons), lengths that are not multiples of 3 or 256 (columns straddling tiles), row
    subsets in any order."""
    monkeypatch.setenv("PFA_TALL_MIN", "0")
    if rows_per_chunk is not None:      # None: the plan's own 48-row chunks (straight-line code) and a shorter last one
        monkeypatch.setenv("PFA_TALL_ROWS", str(rows_per_chunk))
    monkeypatch.setenv("PFA_TALL_SLOTS", str(slots))
    rng = np.random.default_rng(500 + seed)
    b = scan.Batch(stride_align=int(rng.choice([16, 128])))
    units = []
    for _ in range(36):
        n = int(rng.choice([1, 2, 3, 7, 9, 16, 17, 31, 33, 50, 64, 65, 100, 130]))
        L = int(rng.choice([3, 5, 255, 256, 257, 258, 513, 768, 769, 770, 1537, 2304]))
        alpha = ALPHABETS[int(rng.integers(0, 4))]
        m = _random_matrix(rng, n, L, alpha, float(rng.choice([0.0, 0.002, 0.01, 0.03, 0.1, 0.3])))
        mid = b.add_matrix(m)
        b.add_unit(mid, cds=True)
        units.append((m, None, True))
        if n >= 3:
            rows = rng.permutation(n)[: int(rng.integers(1, n + 1))].astype(np.uint32)
            b.add_unit(mid, rows=rows, cds=True)
            units.append((m, rows, True))
    plan = scan.Plan(b.pack()[1])
    assert 0 < plan.rows_per_chunk <= max(rows_per_chunk or 48, 8)
    plan.close()
    out = scan.scan_batch(b, want_pos=True, want_sn=True)
    for i, (m, rows, cds) in enumerate(units):
        _compare(out, i, c_oracle.scan(m, rows=rows, cds=True, want_sites=True), True)


@pytest.mark.parametrize("seed,cds", [(0, False), (1, True), (2, True)])
def test_random_many_row_batches_lean_kernels(gpu, seed, cds):
    """Many-row plans without row subsets and bitmaps: the lean instantiations of the row-chunked
    kernel (what `--dir` batches of large alignments run), coding and not."""
    rng = np.random.default_rng(4242 + seed)
    b = scan.Batch(stride_align=[128, 16, 64][seed])
    units = []
    for _ in range(30):
        n = int(rng.choice([65, 100, 128, 129, 200, 255, 300, 700]))
        L = int(rng.choice([3, 255, 256, 257, 770, 1537, 3001]))
        m = _random_matrix(rng, n, L, ALPHABETS[int(rng.integers(0, 4))], float(rng.choice([0.0, 0.002, 0.01, 0.05])))
        b.add_unit(b.add_matrix(m), cds=cds)
        units.append(m)
    plan = scan.Plan(b.pack()[1])
    assert "scan_tall_kernel<cds=%d,mapped=0,bitmaps=0" % int(cds) in plan.description, plan.description
    plan.close()
    out = scan.scan_batch(b)
    from polyfasta_amd import stats
    for i, m in enumerate(units):
        o = c_oracle.scan(m, cds=cds)
        r = out.results[i]
        assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"]), (i, m.shape)
        assert out.sfs(i).tolist() == o["sfs"].tolist()
        if cds:
            assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
                   (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"]), (i, m.shape)
            assert abs(stats.syn_sites_from_bins(out.syn_num[i]) - o["count_syn"]) <= 1e-9 * max(1.0, o["count_syn"])
