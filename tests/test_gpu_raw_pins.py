"""HIP path vs the REFERENCE's raw counters — no oracle in between.

tests/golden/raw/*.u64.gz hold `bases[]` exactly as the unmodified reference
binary's read_fastq left it (quack.c:223-226; made by oracle/make_raw_goldens.sh
with the LD_PRELOAD observer oracle/ref_peek.c).  Every golden case goes through
the product's host feed + HIP kernels (quack_amd.read_fastq == qkh_accumulate_file)
and must reproduce every cell: all 91 score bins, content, length_count and
kmer_count — incl. bases[10].kmer_count without -a (quack.c:215), which no byte
of the SVG depends on."""
import numpy as np
import pytest

import cases
import quack_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,argv", cases.load(), ids=[c[0] for c in cases.load()])
def test_hip_against_reference_raw_counters(name, argv):
    opt = cases.options(argv)
    kmers = quack_amd.read_adapters(cases.inp(opt["a"])) if "a" in opt else None
    files = [opt["1"], opt["2"]] if ("1" in opt and "2" in opt) else [opt["u"]]
    for panel, f in enumerate(files):
        sd = quack_amd.read_fastq(cases.inp(f), kmers)
        want = cases.raw_table(name, panel)
        assert sd.max_length == want.shape[0]
        assert sd.number_of_sequences == int(want[:, 95].sum())
        if not np.array_equal(sd.bases, want):
            pos, row = np.argwhere(sd.bases != want)[0]
            raise AssertionError("%s panel %d: position %d row %d: hip %d reference %d (%d cells differ)" % (
                name, panel, pos, row, sd.bases[pos, row], want[pos, row], (sd.bases != want).sum()))


def test_sharded_over_three_accumulators_on_one_device():
    """the same through the sharded path (QUACK_DEVICES-style device list 0,0,0): batch round-robin + table sum"""
    for name in ("ragged100_adapters", "long40_adapters", "badcrc_first_of_two"):
        opt = cases.options(dict(cases.load())[name])
        kmers = quack_amd.read_adapters(cases.inp(opt["a"]))
        sd = quack_amd.read_fastq(cases.inp(opt["u"]), kmers, devices=(0, 0, 0))
        assert np.array_equal(sd.bases, cases.raw_table(name))
