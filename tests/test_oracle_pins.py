"""Pin the oracle (oracle/quack_oracle.c) to the reference.

1. Known-answer case verified on the reference during the survey.
2. Integer quantities recovered from SVGs written by the reference's own
   binary (tests/golden/svg, made by oracle/make_goldens.sh) — no renderer of
   ours is involved: svg_counters.parse reads the reference's bytes.
"""
import numpy as np
import pytest

import cases
import oracle_binding as ob
import svg_counters


def test_kat_with_adapters(inputs):
    k = ob.kmers_from_seqs(["ACGTTGCAAGGCT"])
    # only windows [1..10],[2..11],[3..12] are inserted
    assert sorted(np.flatnonzero(k).tolist()) == [385273, 744975, 882750]
    bases, n = ob.read_fastq(cases.inp("kat.fq"), k)
    assert (n, bases.shape[0]) == (6, 14)
    kc = bases[:, 96]
    assert kc[10] == 1 and kc[12] == 1 and kc.sum() == 2
    lc = bases[:, 95]
    assert {int(i): int(lc[i]) for i in np.flatnonzero(lc)} == {2: 1, 9: 1, 11: 2, 13: 2}
    assert bases[0, 91] == 4 and bases[0, 93] == 2          # pos0: A=4 (N->A), C=2
    assert {int(i): int(bases[0, i]) for i in np.flatnonzero(bases[0, :91])} == {0: 1, 2: 1, 20: 3, 40: 1}


def test_kat_without_adapters(inputs):
    bases, n = ob.read_fastq(cases.inp("kat.fq"), None)
    assert n == 6
    assert bases[10, 96] == 4 and bases[:, 96].sum() == 4    # every read longer than 10


def test_bin_rules_match_reference_on_defined_domain():
    lookup = [0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]   # quack.c:150
    for c in list(range(65, 85)) + list(range(97, 117)):
        assert ob.base_code(c) == lookup[(c - 65) & ~32], chr(c)
    for b in range(33, 124):
        assert ob.qual_bin(b) == b - 33
    for b in list(range(0, 33)) + [124, 125, 126, 127]:
        assert ob.qual_bin(b) == -1


def _tables_for(argv):
    opt = cases.options(argv)
    k = ob.kmers_from_file(cases.inp(opt["a"])) if "a" in opt else None
    paired = "1" in opt and "2" in opt
    files = [opt["1"], opt["2"]] if paired else [opt["u"]]
    return [ob.read_fastq(cases.inp(f), k) for f in files], "a" in opt


@pytest.mark.parametrize("name,argv", cases.load(), ids=[c[0] for c in cases.load()])
def test_oracle_against_reference_svg(name, argv):
    svg = cases.golden_svg(name)
    tables, adapters = _tables_for(argv)
    for panel, (bases, n_reads) in enumerate(tables):
        got = svg_counters.parse(svg, panel)
        assert got["n_reads"] == n_reads
        assert got["has_adapters"] == adapters
        if name in cases.BINNED:
            assert got["max_len"] == (bases.shape[0] - 1) // 100     # quack.c:240-261
            continue
        assert got["max_len"] == bases.shape[0]
        want = svg_counters.derive_from_counters(bases, n_reads)
        np.testing.assert_array_equal(got["content"], want["content"])           # raw counts, exact
        np.testing.assert_array_equal(got["length_pct"], want["length_pct"])
        if adapters:
            np.testing.assert_array_equal(got["kmer_cum_pct"], want["kmer_cum_pct"])
        off = 0 if got["encoding"] == "phred33" else 31
        hi = off + got["max_score"]
        np.testing.assert_array_equal(got["score_pct"][:, off:hi], want["score_pct"][:, off:hi])
        if name in cases.EXACT_100:
            # n = score_sum = 100: percentages are the counters themselves
            assert n_reads == 100
            raw = bases.astype(np.int64)
            np.testing.assert_array_equal(got["score_pct"][:, off:hi], raw[:, off:hi])
            np.testing.assert_array_equal(got["length_pct"], raw[:, 95])
            if adapters:
                np.testing.assert_array_equal(got["kmer_cum_pct"], np.cumsum(raw[:, 96]))
            # the only bins draw() does not show must hold the remainder
            assert (raw[:, :91].sum(axis=1) == 100).all()


@pytest.mark.parametrize("name,argv", cases.load(), ids=[c[0] for c in cases.load()])
def test_oracle_against_reference_raw_counters(name, argv):
    """every counter of every position, incl. what draw() never shows: bases[10].kmer_count without -a
    (quack.c:215, kmers == NULL) and the score bins outside the drawn range (quack.c:326-332)"""
    tables, _ = _tables_for(argv)
    for panel, (bases, n_reads) in enumerate(tables):
        want = cases.raw_table(name, panel)
        assert bases.shape == want.shape
        np.testing.assert_array_equal(bases, want)
        assert int(want[:, 95].sum()) == n_reads          # every read has a last base (no empty reads in the goldens)


def test_the_undrawn_kmer_counter_is_the_references():
    """without -a every read longer than 10 counts at bases[10].kmer_count (quack.c:206-217 with kmers == NULL):
    read off the reference's own table"""
    raw = cases.raw_table("uniform100")
    assert raw[10, 96] == 100 and raw[:, 96].sum() == 100
    raw = cases.raw_table("kat")
    assert raw[10, 96] == 4 and raw[:, 96].sum() == 4


def test_tokenizer_corner_cases_agree_with_plain_layout():
    """crlf / truncated / no-final-newline files carry the same 100 records:
    the reference draws identical SVGs for them, so must the oracle's tables."""
    base, n = ob.read_fastq(cases.inp("nonewline100.fq"))
    for other in ("crlf100.fq", "truncated100.fq"):
        b, m = ob.read_fastq(cases.inp(other))
        assert m == n == 100
        np.testing.assert_array_equal(b, base)
    assert cases.golden_svg("crlf100") == cases.golden_svg("nonewline100") == cases.golden_svg("truncated100")
