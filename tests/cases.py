"""Golden case manifest (tests/golden/cases.tsv) shared by several tests."""
import gzip
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
FLAGS = {"-u": "u", "--unpaired": "u", "-1": "1", "--forward": "1", "-2": "2", "--reverse": "2",
         "-a": "a", "--adapters": "a", "-n": "n", "--name": "n"}


def load(manifest="cases.tsv"):
    out = []
    for line in open(os.path.join(G, manifest)):
        if line.startswith("#") or not line.strip():
            continue
        name, _, args = line.rstrip("\n").partition("\t")
        out.append((name, args.split()))
    return out


def options(argv):
    opt = {}
    for i in range(0, len(argv) - 1, 2):
        opt[FLAGS[argv[i]]] = argv[i + 1]
    return opt


def golden_svg(name):
    with gzip.open(os.path.join(G, "svg", name + ".svg.gz")) as f:
        return f.read()


def golden_err(name):
    with open(os.path.join(G, "svg", name + ".err"), "rb") as f:
        return f.read()


def inp(name):
    return os.path.join(G, "inputs", name)


# fixtures with 100 equal-length reads and only valid qualities: every raw
# counter is readable from the reference's SVG
EXACT_100 = {"uniform100", "uniform100_gz", "uniform100_named", "uniform100_adapters",
             "uniform100_longflags", "adapter100", "adapter100_noadapters", "crlf100", "multiline100",
             "truncated100", "nonewline100", "phred64_100", "unpaired_wins_over_half_pair"}
BINNED = {"long40", "long40_adapters"}


def raw_table(name, panel=0):
    """the REFERENCE's raw `bases[]` (quack.c:223-226) for a golden case: [max_length][97] u64, made by
    oracle/make_raw_goldens.sh from the unmodified reference binary (LD_PRELOAD observer oracle/ref_peek.c)"""
    import numpy as np
    with gzip.open(os.path.join(G, "raw", "%s.%d.u64.gz" % (name, panel))) as f:
        return np.frombuffer(f.read(), dtype="<u8").reshape(-1, 97)


# cases whose inputs leave quack.c's defined domain (a quality byte above '{' indexes past scores[91],
# quack.c:203-204: the reference's table is then whatever the aliasing did); none today
RAW_UNDEFINED = set()
