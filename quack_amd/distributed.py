"""One process per GPU: merging the per-rank counter tables.

The accumulation shards by read batch with no data-path collective (every
counter update is a commutative integer ++, quack.c:202,204,216,219,220).  At
the end there is exactly one exchange: all-reduce(MAX) of the geometry (longest
read, table length: 2 words) and ONE all-reduce(SUM) of the integer tables — `ncclAllReduce` over
xGMI when the process group is "nccl" (= RCCL on ROCm), gloo in the CPU tests.
Integer sums are order-independent, so the result is bit-exact whatever
algorithm the backend picks.
"""
import torch
import torch.distributed as dist

from ._capi import QK_N_ROWS


def allreduce_planar(planar, table_len, group=None):
    """planar: int64 tensor of QK_N_ROWS*table_len + 1 words (rows x positions,
    then number_of_sequences).  Returns (summed tensor, common table_len)."""
    if planar.dtype != torch.int64 or planar.numel() != QK_N_ROWS * table_len + 1:
        raise ValueError("planar table must be int64[%d*table_len+1]" % QK_N_ROWS)
    tl = torch.tensor([table_len], dtype=torch.int64, device=planar.device)
    dist.all_reduce(tl, op=dist.ReduceOp.MAX, group=group)
    common = int(tl.item())
    if common != table_len:
        wide = torch.zeros(QK_N_ROWS * common + 1, dtype=torch.int64, device=planar.device)
        wide[:QK_N_ROWS * common].view(QK_N_ROWS, common)[:, :table_len] = \
            planar[:QK_N_ROWS * table_len].view(QK_N_ROWS, table_len)
        wide[-1] = planar[-1]
        planar = wide
    dist.all_reduce(planar, op=dist.ReduceOp.SUM, group=group)
    return planar, common


def allreduce_accumulators(accs, group=None, via_host=False):
    """Sum every rank's qk_accum with its counterparts on the other ranks, in place (GPU path) — for ALL the
    accumulators of `accs` with one exchange: one all-reduce(MAX) of the geometries (two words per accumulator),
    then ONE all-reduce(SUM) over their tables laid end to end (a pair's two mates travel together: quack.c:911-921
    are two independent accumulations, their counters never mix — they only share the message).
    via_host=True stages the tables through host memory for backends without device collectives (gloo; used to
    rehearse the N>1 path on a one-GPU box)."""
    accs = list(accs)
    if not accs:
        return accs
    dev = accs[0].device
    if any(a.device != dev for a in accs):
        raise ValueError("the accumulators of one exchange live on one device (one process per GPU)")
    gpu = torch.device("cuda", dev)
    coll = torch.device("cpu") if via_host else gpu
    # common geometry first, so that every rank exports the same number of words per accumulator
    geo = []
    for a in accs:
        geo += [a.stats()[0], a.table_words()]
    geo = torch.tensor(geo, dtype=torch.int64, device=coll)
    dist.all_reduce(geo, op=dist.ReduceOp.MAX, group=group)
    geo = geo.tolist()
    words = []
    for i, a in enumerate(accs):
        table_len = (geo[2 * i + 1] - 1) // QK_N_ROWS
        a.reserve(table_len)
        assert a.table_words() == QK_N_ROWS * table_len + 1
        words.append(a.table_words())
    buf = torch.empty(sum(words), dtype=torch.int64, device=gpu)
    torch.cuda.current_stream(gpu).synchronize()   # buf is allocated before the shim's streams write it
    at = 0
    for a, w in zip(accs, words):
        a.export_table(buf[at:at + w])             # synchronises the shim's stream before returning
        at += w
    if via_host:
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    # a finished NCCL call only orders torch's stream; the shim imports on its own stream
    torch.cuda.current_stream(gpu).synchronize()
    at = 0
    for i, (a, w) in enumerate(zip(accs, words)):
        a.import_table(buf[at:at + w], geo[2 * i])
        at += w
    return accs


def allreduce_accumulator(acc, group=None, via_host=False):
    """one accumulator: see allreduce_accumulators"""
    return allreduce_accumulators([acc], group=group, via_host=via_host)[0]


def planar_from_bases(bases, number_of_sequences, table_len=None):
    """[max_len, 97] reference layout -> planar int64 tensor (CPU helper)."""
    import numpy as np
    ml = bases.shape[0]
    tl = table_len or max(ml, 1)
    out = np.zeros(QK_N_ROWS * tl + 1, dtype=np.int64)
    out[:QK_N_ROWS * tl].reshape(QK_N_ROWS, tl)[:, :ml] = bases.astype(np.int64).T
    out[-1] = number_of_sequences
    return torch.from_numpy(out), tl


def bases_from_planar(planar, table_len, max_len):
    import numpy as np
    a = planar[:QK_N_ROWS * table_len].view(QK_N_ROWS, table_len)[:, :max_len].cpu().numpy()
    return np.ascontiguousarray(a.T).astype(np.uint64), int(planar[-1].item())
