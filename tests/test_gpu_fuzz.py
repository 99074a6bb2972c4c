"""A seeded, time-boxed stretch of tools/fuzz_gpu.py inside the -m gpu suite (the fuzzer was a
builder-run tool only): random batches of every family — fixed length (packed, padded), packed ragged, long reads on
cache lines, fixed stride + lengths — random shapes, alphabets up to all 256 byte values, adapters loaded and spliced in
or not, host and device routes; every counter against the oracle.  Two seeds: one fixed (reproducible), one from the
date (new ground on every run; the seed is in the failure message)."""
import os
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("seed", [20260104, None], ids=["fixed-seed", "seed-of-the-day"])
def test_random_batches_against_the_oracle(seed):
    import fuzz_gpu
    if seed is None:
        seed = int(time.strftime("%Y%m%d")) * 1000
    done, bad = fuzz_gpu.fuzz(seconds=25, seed=seed, log=lambda m: None)
    assert bad is None, bad
    assert done >= 20, "only %d rounds in 25 s" % done
