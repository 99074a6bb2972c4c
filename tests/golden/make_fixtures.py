#!/usr/bin/env python3
"""Generate the small FASTQ/FASTA fixtures under tests/golden/inputs/.

Deterministic (seeded); the outputs are committed, this script documents how
they were made.  Every input stays inside the reference's *defined* behaviour
(letters A..T/a..t, quality bytes 33..123, read length >= 1) except where a
file name says otherwise, because goldens are produced by the reference
itself (oracle/make_goldens.sh).

100-read, equal-length fixtures are special: transform() turns counts into
`100*c/score_sum` and `ceil(100*c/n)`; with n = score_sum = 100 those are the
raw counts, so tests can recover every integer counter from the reference SVG.
"""
import gzip
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "inputs")

ADAPTERS = [
    ("synthA", "TTGACCGTAGGCATCGGATCCAGTTCAGGACTAGCATG"),
    ("synthB", "GGCATTCAGCTAGGCTTACGGATACCGATGCATCGAGGTACC"),
    ("synthC_short", "ACGTACGTAC"),           # exactly 10 nt: inserts nothing
    ("synthD_13", "ACGTTGCAAGGCT"),            # 3 windows inserted
    ("synthE_N", "CCGTANNTGGCATCGATTGCAGGCTA"),  # N counts as A in the index
]


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def rand_qual(rng, n, lo=2, hi=41, offset=33, levels=None):
    if levels:
        return "".join(chr(offset + rng.choice(levels)) for _ in range(n))
    return "".join(chr(offset + rng.randint(lo, hi)) for _ in range(n))


def write(name, text, gz=False, members=1, newline="\n"):
    data = text.replace("\n", newline).encode("ascii")
    path = os.path.join(OUT, name)
    if gz:
        # mtime=0 keeps the bytes reproducible
        if members == 1:
            with open(path, "wb") as f, gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
                g.write(data)
        else:
            cut = [len(data) * i // members for i in range(members + 1)]
            with open(path, "wb") as f:
                for i in range(members):
                    with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
                        g.write(data[cut[i]:cut[i + 1]])
    else:
        with open(path, "wb") as f:
            f.write(data)


def fastq(records, plus_repeat=False):
    out = []
    for name, s, q in records:
        out.append("@%s\n%s\n+%s\n%s\n" % (name, s, name if plus_repeat else "", q))
    return "".join(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = random.Random(20261003)
    levels8 = [2, 11, 14, 22, 27, 33, 37, 40]

    # adapters FASTA (own synthetic set, multi-line record + lowercase included)
    fa = []
    for name, s in ADAPTERS:
        if name == "synthB":
            fa.append(">%s second line wrapped\n%s\n%s\n" % (name, s[:20], s[20:].lower()))
        else:
            fa.append(">%s\n%s\n" % (name, s))
    write("adapters.fa", "".join(fa))
    write("adapters.fa.gz", "".join(fa), gz=True)

    # 1. SURVEY 8c known-answer reads (without the Q=93 read: undefined there)
    kat = [("r0", "ACGTTGCAAGAAAA", "I" * 14), ("r1", "CGTTGCAAGGAA", "#" * 12),
           ("r2", "AACGTTGCAAGGAA", "5" * 14), ("r3", "AACGTTGCAAGG", "5" * 12),
           ("r4", "CGTTGCAAGG", "5" * 10), ("r5", "Ngt", "!+J")]
    write("kat.fq", fastq(kat))
    write("kat_adapter.fa", ">ad\nACGTTGCAAGGCT\n")

    # 2. 100 x 60 bp, 8 quality levels, some N / lowercase / IUPAC letters
    recs = []
    for i in range(100):
        s = list(rand_seq(rng, 60))
        for _ in range(3):
            s[rng.randrange(60)] = rng.choice("NnacgtRKMSBDH")
        recs.append(("u%d comment here" % i, "".join(s), rand_qual(rng, 60, levels=levels8)))
    write("uniform100.fq", fastq(recs))
    write("uniform100.fq.gz", fastq(recs), gz=True)

    # 3. 100 reads with adapters spliced in (length 80), exercises first-hit,
    #    hit-ending-on-last-base, hit-in-first-window and no-hit paths
    recs = []
    for i in range(100):
        s = rand_seq(rng, 80)
        mode = i % 5
        ad = ADAPTERS[i % 2][1]
        if mode == 0:
            at = rng.randrange(0, 60)
            s = (s[:at] + ad)[:80]
            s = s + rand_seq(rng, 80 - len(s))
        elif mode == 1:                       # adapter tail cut by the read end
            at = rng.randrange(62, 75)
            s = (s[:at] + ad)[:80]
        elif mode == 2:                       # read starts inside the adapter
            s = (ad[3:] + s)[:80]
        elif mode == 3:                       # a window that ends exactly on the last base
            s = s[:69] + ad[1:12]
        recs.append(("a%d" % i, s, rand_qual(rng, 80, levels=levels8)))
    write("adapter100.fq", fastq(recs))

    # 4. ragged lengths 1..90 (two reads shorter than 10, one of length 10)
    recs = []
    for i in range(100):
        n = [1, 3, 10, 11, 90][i] if i < 5 else rng.randint(12, 90)
        recs.append(("g%d" % i, rand_seq(rng, n), rand_qual(rng, n, levels=levels8)))
    write("ragged100.fq", fastq(recs))
    write("ragged100_2member.fq.gz", fastq(recs), gz=True, members=2)

    # 5. tokenizer corner cases (same 100 x 60 records as fixture 2 re-wrapped)
    base = [(n.split()[0], s, q) for n, s, q in
            [("u%d" % i, rand_seq(rng, 60), rand_qual(rng, 60, levels=levels8)) for i in range(100)]]
    write("crlf100.fq", fastq(base), newline="\r\n")
    ml = []
    for k, (n, s, q) in enumerate(base):
        # multi-line FASTQ: sequence over three lines with a blank line inside,
        # '+name' repeated, quality over two lines whose first bytes may be '@'
        # or '+' (kseq reads quality by length, so that is legal)
        q = list(q)
        if k % 2 == 0:
            q[0] = "@" if k % 4 == 0 else "+"
        if k % 3 == 0:
            q[25] = "@"
        q = "".join(q)
        ml.append("@%s desc\n%s\n\n%s\n%s\n+%s\n%s\n%s\n"
                  % (n, s[:20], s[20:45], s[45:], n, q[:25], q[25:]))
    write("multiline100.fq", "".join(ml))
    # truncated final record: 100 complete + a header/sequence with no quality
    write("truncated100.fq", fastq(base) + "@cut\nACGTACGTAC\n+\nIIII")
    # no trailing newline on the last quality line
    write("nonewline100.fq", fastq(base)[:-1])
    # phred64: every quality byte >= 64 (';' .. 'h' style), scores 0..40 + 64
    p64 = [("p%d" % i, rand_seq(rng, 60), rand_qual(rng, 60, lo=2, hi=40, offset=64)) for i in range(100)]
    write("phred64_100.fq", fastq(p64))

    # 6. paired mates (R2 skewed lower), 100 x 75
    r1 = [("m%d/1" % i, rand_seq(rng, 75), rand_qual(rng, 75, levels=levels8)) for i in range(100)]
    r2 = [("m%d/2" % i, rand_seq(rng, 75), rand_qual(rng, 75, levels=levels8[:5])) for i in range(100)]
    write("paired_R1.fq.gz", fastq(r1), gz=True)
    write("paired_R2.fq.gz", fastq(r2), gz=True)

    # 7. long reads: max_length > 3000 -> "Binning..." (x100), 40 reads 800..5200
    recs = []
    for i in range(40):
        n = 5200 if i == 0 else rng.randint(800, 5000)
        recs.append(("L%d" % i, rand_seq(rng, n), rand_qual(rng, n, lo=1, hi=60)))
    write("long40.fq.gz", fastq(recs), gz=True)

    # 8. exactly 500 positions (the largest the reference's averages[500] holds)
    recs = [("w%d" % i, rand_seq(rng, 500 if i == 0 else rng.randint(100, 500)),
             None) for i in range(30)]
    recs = [(n, s, rand_qual(rng, len(s), levels=levels8[2:6])) for n, s, _ in recs]
    write("len500.fq.gz", fastq(recs), gz=True)

    # 9. a single read, and a FASTQ whose reads are all 1 base long
    write("single_read.fq", fastq([("only", "ACGTNACGTTGCA", "IIIIIHHHHH###")]))
    write("one_base.fq", fastq([("b%d" % i, "ACGTN"[i % 5], "5") for i in range(20)]))

    # 10. gzip members whose trailer does not match their data (a flipped bit in the stored CRC-32, a wrong
    #     ISIZE): zlib's gzread fails the 16 KiB read in which inflate meets the trailer, so the reference
    #     loses the bytes from the last multiple of 16384 on and stops there
    recs = [("x%d" % i, rand_seq(rng, 100), rand_qual(rng, 100, levels=levels8[3:7])) for i in range(800)]
    text = fastq(recs)
    write("good800.fq.gz", text, gz=True)
    good = open(os.path.join(OUT, "good800.fq.gz"), "rb").read()
    os.remove(os.path.join(OUT, "good800.fq.gz"))
    bad = bytearray(good)
    bad[-6] ^= 0x10                               # CRC-32 field = bytes -8..-5
    open(os.path.join(OUT, "badcrc800.fq.gz"), "wb").write(bytes(bad))
    bad = bytearray(good)
    bad[-4] ^= 0x01                               # ISIZE field = bytes -4..-1
    open(os.path.join(OUT, "badlen800.fq.gz"), "wb").write(bytes(bad))
    # two members, the FIRST one damaged: nothing of the second is ever read
    write("two_members.tmp.gz", text, gz=True, members=2)
    two = bytearray(open(os.path.join(OUT, "two_members.tmp.gz"), "rb").read())
    os.remove(os.path.join(OUT, "two_members.tmp.gz"))
    second = two.index(b"\x1f\x8b\x08", 10)      # start of the second member
    two[second - 7] ^= 0x80
    open(os.path.join(OUT, "badcrc_first_of_two.fq.gz"), "wb").write(bytes(two))


if __name__ == "__main__":
    main()
