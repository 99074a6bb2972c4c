"""Recover integer quantities from an SVG written by the reference (or by the
drop-in CLI).  The reference never prints its raw counters, but draw() leaks:
  - number_of_sequences and max_length                    quack.c:354,418
  - cumulative RAW base-content counts per position       quack.c:455-463
  - 100*score/score_sum per (position, score)             quack.c:285,573
  - ceil(100*length_count/n), ceil(100*cum_kmer/n)        quack.c:288-289,657,718
With 100 equal-length reads the percentages ARE the counts."""
import re

import numpy as np


def parse(svg_bytes, panel=0):
    text = svg_bytes.decode("ascii")
    stats = list(re.finditer(r"<tspan>\n(-?\d+)\s*</tspan>(?:.|\n)*?<tspan>\n(phred\d\d)", text))
    n_reads = int(stats[panel].group(1))
    encoding = stats[panel].group(2)
    # split the document into per-panel chunks at the file-stats <text>
    starts = [m.start() for m in stats] + [len(text)]
    chunk = text[starts[panel]:starts[panel + 1]]
    vb = re.search(r'height="100" preserveAspectRatio="none" viewBox="0 0 (\d+) (-?\d+)"', chunk)
    max_len = int(vb.group(1))
    assert int(vb.group(2)) == n_reads
    # base content: polylines are emitted G,C,T,A (stack top first); points
    # "0,0 0,y0 0.5,y ... L,y L,0"
    polys = re.findall(r'<polyline points="0,0 ([^"]*?) \d+,0" fill="(#[0-9a-f]{6})" stroke="none"/>', chunk)
    assert len(polys) == 4
    order = {"#648964": 0, "#89bc89": 1, "#84accf": 2, "#5d7992": 3}
    cum = np.zeros((max_len, 4), dtype=np.int64)
    for pts, color in polys:
        i = order[color]
        for tok in pts.split():
            xs, ys = tok.split(",")
            if xs.endswith(".5"):
                cum[int(xs[:-2]), i] = int(ys)
    content = np.diff(np.concatenate([np.zeros((max_len, 1), np.int64), cum], axis=1), axis=1)
    # heat map
    hm = re.search(r'height="250" preserveAspectRatio="none" viewBox="0 0 \d+ (\d+)"', chunk)
    max_score = int(hm.group(1))
    offset = 0 if encoding == "phred33" else 31
    pct = np.zeros((max_len, 91), dtype=np.int64)
    for m in re.finditer(r'<rect x="(\d+)" y="(\d+)" fill-opacity="([0-9.]+)" width="1" height="1"', chunk):
        pct[int(m.group(1)), int(m.group(2)) + offset] = int(round(float(m.group(3)) * 100))
    means = re.search(r'<polyline points="(0,[^"]*)" stroke="black"', chunk).group(1)

    def bars(y0):
        m = re.search(r'<svg x="0" y="%d" width="450" height="100"[^>]*>(.*?)</svg>' % y0, chunk, re.S)
        out = np.zeros(max_len, dtype=np.int64)
        if m:
            for r in re.finditer(r'<rect x="(\d+)" y="0" width="1" height="(-?\d+)"', m.group(1)):
                out[int(r.group(1))] = int(r.group(2))
        return out, m is not None

    length_pct, _ = bars(360)
    kmer_pct, has_adapters = bars(465)
    return dict(n_reads=n_reads, max_len=max_len, encoding=encoding, content=content, score_pct=pct,
                max_score=max_score, length_pct=length_pct, kmer_cum_pct=kmer_pct,
                has_adapters=has_adapters, means=means)


def derive_from_counters(bases, n_reads):
    """The same quantities computed from raw counters with transform()'s
    formulas (quack.c:264-291), for tables that are not binned."""
    bases = bases.astype(np.int64)
    scores = bases[:, :91]
    ssum = scores.sum(axis=1)
    pct = np.where(ssum[:, None] != 0, 100 * scores // np.maximum(ssum[:, None], 1), scores)
    f32 = np.float32
    length_pct = np.ceil((f32(100) * bases[:, 95].astype(f32) / f32(n_reads)).astype(np.float64)).astype(np.int64)
    cum = np.cumsum(bases[:, 96])
    kmer_pct = np.ceil((f32(100) * cum.astype(f32) / f32(n_reads)).astype(np.float64)).astype(np.int64)
    return dict(content=bases[:, 91:95], score_pct=pct, length_pct=length_pct, kmer_cum_pct=kmer_pct)
