"""N>1 path on CPU: two processes (gloo), each holding the table of its shard
of the read batches; one all-reduce(MAX) of the table length and ONE
all-reduce(SUM) of the integer tables must equal the single-process table."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle_binding as ob
from quack_amd import distributed as qd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_tables(path, world, kmers):
    """deal the file's reads round-robin in groups of 7 (a 'batch') to ranks"""
    recs, _ = ob.tokenize(path)
    shards = [[] for _ in range(world)]
    for i in range(0, len(recs), 7):
        shards[(i // 7) % world].extend(recs[i:i + 7])
    out = []
    for sh in shards:
        seq = np.frombuffer(b"".join(s for s, _ in sh), np.uint8)
        qual = np.frombuffer(b"".join(q for _, q in sh), np.uint8)
        off = np.concatenate([[0], np.cumsum([len(s) for s, _ in sh])]).astype(np.uint64)
        out.append(ob.accumulate_batch(seq, qual, off, kmers=kmers))
    return out


def _worker(rank, world, port, path, adapters, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kmers = ob.kmers_from_file(adapters) if adapters else None
        bases, n = _shard_tables(path, world, kmers)[rank]
        planar, tl = qd.planar_from_bases(bases, n)          # ranks have different table lengths
        ml = torch.tensor([bases.shape[0]])
        dist.all_reduce(ml, op=dist.ReduceOp.MAX)
        summed, tl = qd.allreduce_planar(planar, tl)
        got, n_all = qd.bases_from_planar(summed, tl, int(ml.item()))
        if rank == 0:
            q.put((got, n_all))
    finally:
        dist.destroy_process_group()


def _run(path, adapters, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, adapters, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, n_all = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return got, n_all


def test_two_ranks_equal_one(inputs):
    path = cases.inp("ragged100.fq")
    want, n = ob.read_fastq(path)
    got, n_all = _run(path, None)
    assert n_all == n == 100
    np.testing.assert_array_equal(got, want)


def test_two_ranks_with_adapters(inputs):
    path, ad = cases.inp("adapter100.fq"), cases.inp("adapters.fa")
    want, n = ob.read_fastq(path, ob.kmers_from_file(ad))
    got, n_all = _run(path, ad)
    assert n_all == n
    np.testing.assert_array_equal(got, want)
