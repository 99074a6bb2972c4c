"""quack_amd — MI355X-native accumulation path for quack (IGBB/quack).

The product is the native code: `libquack_hip.so` (hand-written gfx950 kernels
behind the C-ABI of include/quack_hip.h) and the C host in quack_amd/host
(tokenizer, CLI, byte-identical SVG).  This package is the Python mirror of
the reference's function-level interface for that path:

    read_adapters(path)           quack.c:154-178
    read_fastq(path, kmers)       quack.c:180-228
    transform / draw              quack.c:230-293, 295-856  (render_svg)

plus `Accumulator`, a thin wrapper over the C-ABI used by bench.py and the
tests, and `distributed`, the one-process-per-GPU merge (RCCL all-reduce of the
integer tables through torch.distributed).
"""
from .api import (Accumulator, HipUnavailable, SequenceData, read_adapters, read_fastq,
                  render_svg, device_count)

__all__ = ["Accumulator", "HipUnavailable", "SequenceData", "read_adapters", "read_fastq",
           "render_svg", "device_count"]
