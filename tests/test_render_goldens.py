"""Host transform()+draw() restatement (quack_amd/host/render.c) must write the
reference's bytes: oracle counters -> qkh_render_document == golden SVG."""
import pytest

import cases
import oracle_binding as ob
import quack_amd


@pytest.mark.parametrize("name,argv", cases.load(), ids=[c[0] for c in cases.load()])
def test_svg_bytes(name, argv):
    opt = cases.options(argv)
    k = ob.kmers_from_file(cases.inp(opt["a"])) if "a" in opt else None
    paired = "1" in opt and "2" in opt
    f = quack_amd.SequenceData(*ob.read_fastq(cases.inp(opt["1"] if paired else opt["u"]), k))
    r = quack_amd.SequenceData(*ob.read_fastq(cases.inp(opt["2"]), k)) if paired else None
    svg, err = quack_amd.render_svg(f, r, name=opt.get("n"), adapters="a" in opt)
    assert err == cases.golden_err(name)           # "Binning...\n" or empty
    assert svg == cases.golden_svg(name)


def test_render_rejects_empty():
    import numpy as np
    with pytest.raises(ValueError):
        quack_amd.render_svg(quack_amd.SequenceData(np.zeros((0, 97), np.uint64), 0))
