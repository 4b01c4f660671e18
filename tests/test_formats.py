"""On-disk formats either side of the hot path: MatrixMarket -> CSR (mtx2csr.cc), CSV writer, binary cache."""
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

import flex_amd
import oracle
from conftest import GOLDEN

MTX = {
    "general_real": """%%MatrixMarket matrix coordinate real general
% a comment
%
4 5 6
1 1 1.5
3 2 -2.25
1 4 3.0e-1
4 5 7
2 3 0.125
1 2 9
""",
    "symmetric_real": """%%MatrixMarket matrix coordinate real symmetric
4 4 5
1 1 2.0
3 1 -1.0
4 2 0.5
4 4 4.0
2 1 3.5
""",
    "pattern_general": """%%MatrixMarket matrix coordinate pattern general
3 3 4
2 1
1 3
3 3
3 1
""",
    "integer_symmetric": """%%MatrixMarket matrix coordinate integer symmetric
3 3 3
2 1 -1
3 3 1
3 2 1
""",
}


@pytest.mark.parametrize("name", sorted(MTX))
def test_mtx_matches_oracle_layout_and_scipy(tmp_path, name):
    path = tmp_path / f"{name}.mtx"
    path.write_text(MTX[name])
    m, n, rp, col, val = oracle.mtx_load(str(path))                  # mtx2csr.cc's unsorted layout
    a = flex_amd.mtx_load(str(path), sort_columns=False)
    assert (a.m, a.n) == (m, n)
    assert np.array_equal(a.rowPtr, rp) and np.array_equal(a.col, col) and np.array_equal(a.vals, val)
    s = flex_amd.mtx_load(str(path), sort_columns=True)
    ref = sp.csr_matrix(scipy.io.mmread(str(path)))                  # independent reader
    ref.sort_indices()
    assert np.array_equal(s.rowPtr, ref.indptr) and np.array_equal(s.col, ref.indices)
    assert np.array_equal(s.vals, ref.data.astype(np.float32))


def test_mtx_errors(tmp_path):
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(flex_amd.FlexError, match="does not parse"):
        flex_amd.mtx_load(str(bad))
    oob = tmp_path / "oob.mtx"
    oob.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n")
    with pytest.raises(flex_amd.FlexError):
        flex_amd.mtx_load(str(oob))
    with pytest.raises(flex_amd.FlexError, match="could not be read"):
        flex_amd.mtx_load(str(tmp_path / "missing.mtx"))
    # found by the sanitizer pass (tools/asan_host.sh): numbers must come from the entry's OWN line, the file may
    # end without a newline, and a header may not promise more entries than the file can hold
    hdr = "%%MatrixMarket matrix coordinate real general\n"
    for name, text in (("short_last.mtx", hdr + "3 3 2\n1 1 1.0\n2"),            # last line: row only, no newline
                       ("short_mid.mtx", hdr + "3 3 2\n1\n2 2 1.0\n"),            # column would be read from the next line
                       ("no_value.mtx", hdr + "3 3 2\n1 1\n2 2 1.0\n"),           # value would be read from the next line
                       ("liar.mtx", hdr + "3 3 4000000000\n1 1 1.0\n"),            # 4e9 entries promised
                       ("huge.mtx", hdr + "3 3 99999999999999999999\n1 1 1.0\n")):
        f = tmp_path / name
        f.write_bytes(text.encode())
        with pytest.raises(flex_amd.FlexError):
            flex_amd.mtx_load(str(f))
    ok = tmp_path / "no_newline.mtx"
    ok.write_bytes((hdr + "2 2 2\n1 1 1.5\n2 2 -2").encode())                    # complete entry, no final newline: fine
    a = flex_amd.mtx_load(str(ok))
    assert a.nnz == 2 and np.array_equal(a.vals, np.array([1.5, -2.0], dtype=np.float32))


def test_conv_binary_is_the_references_mtx2csr(tmp_path):
    """flex_amd/lib/conv in.mtx out.csv (≙ prepare_mtx_data.sh's `conv`): the CSV it writes loads back to the CSR the
    library reads from the MatrixMarket file directly; bad input exits non-zero."""
    import subprocess
    exe = os.path.join(os.path.dirname(flex_amd.lib_path()), "conv")
    assert os.path.exists(exe)
    src = tmp_path / "g.mtx"
    src.write_text("%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 2.0\n2 1 -1.5\n3 3 1.0\n4 2 0.25\n4 4 3.0\n")
    out = subprocess.run([exe, str(src), str(tmp_path / "g.csv"), "--sort"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "m = 4,   n = 4" in out.stdout, out.stdout + out.stderr
    a, b = flex_amd.csv_load(str(tmp_path / "g.csv")), flex_amd.mtx_load(str(src), sort_columns=True)
    assert np.array_equal(a.rowPtr, b.rowPtr) and np.array_equal(a.col, b.col) and np.array_equal(a.vals, b.vals)
    assert a.nnz == 7  # the two off-diagonal entries are mirrored
    bad = subprocess.run([exe, str(tmp_path / "missing.mtx"), str(tmp_path / "x.csv")], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "could not be read" in bad.stderr


def test_mtx_to_csv_is_what_dataloader_reads(tmp_path):
    """prepare_mtx_data.sh: mtx -> conv -> csv -> DataLoader.  Same pipeline, exact fp32 round trip."""
    path = tmp_path / "g.mtx"
    path.write_text(MTX["symmetric_real"])
    a = flex_amd.mtx_load(str(path))
    flex_amd.csv_save(str(tmp_path / "g.csv"), a)
    b = flex_amd.csv_load(str(tmp_path / "g.csv"))
    o = oracle.csv_load(str(tmp_path / "g.csv"))
    for x in (b, o):
        assert np.array_equal(x.rowPtr, a.rowPtr) and np.array_equal(x.col, a.col) and np.array_equal(x.vals, a.vals)
    assert not b.is_directed and b.n_edges_asymmetric == 0


def test_csv_and_binary_round_trip_pubmed(tmp_path):
    a = flex_amd.csv_load(os.path.join(GOLDEN, "pubmed.csv"))
    flex_amd.csv_save(str(tmp_path / "p.csv"), a)
    flex_amd.csr_save_bin(str(tmp_path / "p.bin"), a)
    b = flex_amd.csv_load(str(tmp_path / "p.csv"))
    c = flex_amd.csr_load_bin(str(tmp_path / "p.bin"))
    for x in (b, c):
        assert np.array_equal(x.rowPtr, a.rowPtr) and np.array_equal(x.col, a.col) and np.array_equal(x.vals, a.vals)
        assert (x.n_edges_one_way, x.is_directed, x.uni_nb) == (a.n_edges_one_way, a.is_directed, a.uni_nb)
    with open(tmp_path / "junk.bin", "wb") as f:
        f.write(b"not a csr file at all")
    with pytest.raises(flex_amd.FlexError, match="does not parse"):
        flex_amd.csr_load_bin(str(tmp_path / "junk.bin"))
    raw = open(tmp_path / "p.bin", "rb").read()
    for name, data in (("short.bin", raw[:-4]), ("long.bin", raw + b"\0\0\0\0"),
                       ("liar.bin", raw[:8] + np.array([a.m, a.n, 2 ** 31], dtype=np.int64).tobytes() + raw[32:])):
        (tmp_path / name).write_bytes(data)  # header and file size must agree before anything is allocated
        with pytest.raises(flex_amd.FlexError, match="does not parse"):
            flex_amd.csr_load_bin(str(tmp_path / name))


def test_permutation_cache_round_trip_and_refusals(tmp_path, golden):
    """flex_perm_save / flex_perm_load: a cached ordering comes back bit-identical, and a file
    that is stale (other matrix), truncated, or not a permutation is refused instead of applied."""
    a = flex_amd.csv_load(os.path.join(GOLDEN, "pubmed.csv"))
    fp = flex_amd.csr_fingerprint(a)
    assert fp != 0 and fp == flex_amd.csr_fingerprint(flex_amd.csv_load(os.path.join(GOLDEN, "pubmed.csv")))
    rank = flex_amd.order_rcm(a)
    path = str(tmp_path / "pubmed.RCM.perm")
    flex_amd.perm_save(path, rank, fp)
    back = flex_amd.perm_load(path, a.n, fp)
    assert np.array_equal(back, rank)
    vo = np.empty(a.n, dtype=np.int64)
    vo[back] = np.arange(a.n)
    assert np.array_equal(vo.astype(np.int32), golden["pubmed_rcm_vo_mp"])  # still the reference's RCM
    # values do not enter the fingerprint (they do not change an ordering); structure does
    a2 = flex_amd.HostCsr(a.rowPtr.copy(), a.col.copy(), a.vals * 2, n=a.n)
    assert flex_amd.csr_fingerprint(a2) == fp
    col2 = a.col.copy()
    col2[0], col2[1] = col2[1], col2[0]
    assert flex_amd.csr_fingerprint(flex_amd.HostCsr(a.rowPtr.copy(), col2, a.vals.copy(), n=a.n)) != fp
    with pytest.raises(flex_amd.FlexError):  # absent
        flex_amd.perm_load(str(tmp_path / "nope.perm"), a.n, fp)
    with pytest.raises(flex_amd.FlexError, match="does not parse"):  # written for another matrix
        flex_amd.perm_load(path, a.n, fp ^ 1)
    with pytest.raises(flex_amd.FlexError, match="does not parse"):  # wrong length
        flex_amd.perm_load(path, a.n - 1, fp)
    raw = open(path, "rb").read()
    open(tmp_path / "short.perm", "wb").write(raw[:-4])
    with pytest.raises(flex_amd.FlexError, match="does not parse"):
        flex_amd.perm_load(str(tmp_path / "short.perm"), a.n, fp)
    dup = rank.copy()
    dup[5] = dup[6]
    flex_amd.perm_save(str(tmp_path / "dup.perm"), dup, fp)
    with pytest.raises(flex_amd.FlexError, match="does not parse"):  # not a permutation
        flex_amd.perm_load(str(tmp_path / "dup.perm"), a.n, fp)


def test_parsers_survive_mutated_inputs(tmp_path):
    """600 seeded mutations (byte flips, truncations, duplicated and deleted spans, huge numbers) of a small CSV and
    a small MatrixMarket file: every load either fails with a FlexError or returns a CSR that passes the
    library's own validation -- never a crash, a hang or an inconsistent structure (tools/asan_host.sh runs
    this under AddressSanitizer / UBSan)."""
    rng = np.random.default_rng(77)
    csv = open(os.path.join(GOLDEN, "a_mat.csv"), "rb").read()
    mtx = (b"%%MatrixMarket matrix coordinate real general\n% comment\n6 6 9\n1 1 1.5\n2 1 -2\n3 3 4e-1\n6 2 1\n"
           b"4 5 2.25\n5 5 1\n1 6 3\n2 2 7\n6 6 -1\n")
    bombs = [b"99999999999999999999", b"-1", b"1e400", b"nan", b"4294967296", b",,,,", b"\n\n\n", b"\x00", b"2147483648 2147483648 1"]

    def mutate(data):
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            op = int(rng.integers(0, 5))
            pos = int(rng.integers(0, max(1, len(d))))
            if op == 0 and d:
                d[pos % len(d)] = int(rng.integers(0, 256))
            elif op == 1:
                d = d[:pos]
            elif op == 2:
                span = d[pos:pos + int(rng.integers(1, 40))]
                d[pos:pos] = span
            elif op == 3:
                del d[pos:pos + int(rng.integers(1, 40))]
            else:
                d[pos:pos] = bombs[int(rng.integers(0, len(bombs)))]
        return bytes(d)

    ok = bad = 0
    for i in range(600):
        is_csv = i % 2 == 0
        path = tmp_path / ("m.csv" if is_csv else "m.mtx")
        path.write_bytes(mutate(csv if is_csv else mtx))
        try:
            a = flex_amd.csv_load(str(path)) if is_csv else flex_amd.mtx_load(str(path), sort_columns=bool(i % 4 == 1))
        except flex_amd.FlexError:
            bad += 1
            continue
        ok += 1
        rp = a.rowPtr.astype(np.int64)
        assert len(rp) == a.m + 1 and rp[0] == 0 and np.all(np.diff(rp) >= 0) and rp[-1] == a.nnz == len(a.col) == len(a.vals)
        assert a.nnz == 0 or int(a.col.max()) < a.n
    assert ok > 20 and bad > 20, (ok, bad)  # the mutations exercise both outcomes
