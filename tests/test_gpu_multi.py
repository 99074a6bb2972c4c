"""The N>1 path on real devices (SURVEY §4 tier iv: 1-GPU == N-GPU == oracle).

Two groups:
  * what a ONE-GPU box can run: ranks that share device 0 (gloo exchange of HIP
    tables), bench.py's self-launch, and the shim's RCCL call sequence forced
    on with a world of one device (QUACK_HIP_RCCL_ALWAYS=1);
  * what needs >= 2 GPUs (skipped, not failed, below that): qk_accum_allreduce
    over distinct devices, torch.distributed "nccl" ranks, the paired CLI with
    QUACK_DEVICES=0..N-1 — config 4's shape (quack.c:911-921).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cases
import oracle_binding as ob
import synth
import quack_amd

pytestmark = pytest.mark.gpu

ROOT = cases.ROOT
QUACK = os.path.join(ROOT, "quack_amd", "host", "quack")


def n_devices():
    try:
        return quack_amd.device_count()
    except Exception:
        return 0


# at most 6 processes may use a card at once on the GPU boxes; 4 ranks are plenty
N_MULTI = min(n_devices(), 4)
need2 = pytest.mark.skipif(N_MULTI < 2, reason="needs >= 2 GPUs")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world, backend, devices, tmp_path):
    out = tmp_path / "rank0.json"
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_worker.py"),
                                       backend, devices, str(out)], env=env))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=300))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world, codes
    return json.load(open(out))


# ------------------------------------------------------------------ one GPU is enough
def test_two_ranks_share_one_gpu_gloo_exchange_of_hip_tables(tmp_path):
    """one process per rank, HIP accumulators, tables merged through
    quack_amd.distributed (host-staged gloo): equals the oracle over all reads"""
    res = run_ranks(2, "gloo", "same", tmp_path)
    assert res["ok"] and res["world"] == 2 and res["reads"] == 24000 and res["kmer_hits"] > 1000


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the
    ranks before touching the GPU and relays rank 0's line (here both ranks on
    device 0, gloo exchange; the driver's run uses nccl on distinct devices)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--device", "0", "--reads", "200000", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and len(line["ranks"]["per_rank"]) == 2
    assert "paired" in line["config"]["workload"]          # N>1 runs config 4's per-GPU share
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1


def test_bench_under_the_drivers_launcher(tmp_path):
    """the driver's own command line for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — here with two ranks on device 0 and a gloo
    exchange: RANK / LOCAL_RANK / WORLD_SIZE come from the launcher; the line carries the like-for-like one-GPU
    reference (measured on rank 0 before the group forms) and config 3's share with its own"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0",
                        "--reads", "200000", "--steps", "3", "--warmup", "1", "--also-steps", "3", "--also-warmup", "2"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["ranks"]["world_size"] == 2
    assert line["n1_reference"]["value"] > 0 and 0 < line["efficiency_vs_n1_reference"] < 1.5
    share = line["also"]["cfg3_share"]
    assert "300 bp" in share["workload"] and share["n1_reference"]["value"] > 0 and len(share["per_rank"]) == 2
    assert "ONE all-reduce" in line["ranks"]["exchange"]
    # round 4: the first thing a multi-rank run does is a correctness run — every rank its own reads (another longest read per
    # rank), one exchange, the result against the oracle over the union
    pc = line["ranks"]["parity_check"]
    assert pc["ok"] is True and pc["reads"] == 3000 + 3400 + 2001 + 2002 and pc["max_length"] == 157 and pc["kmer_hits"] > 500


def test_bench_fails_when_a_rank_fails():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--device", "99", "--reads", "1000", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rccl_call_sequence_on_one_device(monkeypatch):
    """qk_accum_allreduce's RCCL branch (dlopen, ncclCommInitAll, grouped
    ncclAllReduce(ncclUint64, ncclSum), cached communicator) with a world of one
    device: the table must come back unchanged, twice (second call = cached comm)"""
    from quack_amd import api
    monkeypatch.setenv("QUACK_HIP_RCCL_ALWAYS", "1")
    seq, qual, off = synth.ragged(20000, 1, 200, seed=91)
    want = ob.accumulate_batch(seq, qual, off)
    accs = [quack_amd.Accumulator(0) for _ in range(2)]
    try:
        h = len(off) // 2
        accs[0].submit(seq[:int(off[h])], qual[:int(off[h])], off[:h + 1])
        accs[1].submit(seq[int(off[h]):], qual[int(off[h]):], off[h:] - off[h])
        api.allreduce(accs)          # add kernel (same device) + RCCL over the one leader
        api.allreduce(accs[:1])      # again: the cached communicator
        sd = accs[0].finish()
        assert sd.number_of_sequences == want[1] and np.array_equal(sd.bases, want[0])
    finally:
        for a in accs:
            a.close()


def test_paired_cli_mates_share_the_rccl_section():
    """both mate threads end in qk_accum_allreduce at about the same time: the RCCL
    section is serialised and its communicator shared (cli.c)"""
    argv = dict(cases.load())["paired_adapters_named"]
    r = subprocess.run([QUACK] + argv, capture_output=True, cwd=cases.inp(""), timeout=240,
                       env=dict(os.environ, QUACK_DEVICES="0,0", QUACK_HIP_RCCL_ALWAYS="1", QUACK_HIP_BATCH_MB="1"))
    assert r.returncode == 0, r.stderr
    assert r.stdout == cases.golden_svg("paired_adapters_named")


# ------------------------------------------------------------------ >= 2 GPUs
@need2
def test_allreduce_over_distinct_devices_equals_one_accumulator_equals_oracle():
    from quack_amd import api
    seq, qual, off = synth.ragged(40000, 1, 300, seed=101)
    k = ob.kmers_from_seqs(synth.synthetic_adapters())
    bits = ob.kmers_to_bitset(k)
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    with quack_amd.Accumulator(0, bits) as one:
        one.submit(seq, qual, off)
        single = one.finish()
    assert single.number_of_sequences == want[1] and np.array_equal(single.bases, want[0])
    accs = [quack_amd.Accumulator(d, bits, max_len_hint=8) for d in range(N_MULTI)]
    try:
        n = len(off) - 1
        cut = [n * i // 13 for i in range(14)]
        for b, (a, e) in enumerate(zip(cut, cut[1:])):
            lo, hi = int(off[a]), int(off[e])
            accs[b % N_MULTI].submit(seq[lo:hi], qual[lo:hi], off[a:e + 1] - off[a])
        api.allreduce(accs)                      # ONE ncclAllReduce(ncclUint64, ncclSum) over xGMI
        for acc in accs:                         # every device now holds the global table
            sd = acc.finish()
            assert sd.number_of_sequences == want[1] and np.array_equal(sd.bases, want[0])
        api.allreduce(accs[:2])                  # a second device set: its own communicators
    finally:
        for acc in accs:
            acc.close()


@need2
def test_nccl_ranks_on_distinct_devices(tmp_path):
    res = run_ranks(N_MULTI, "nccl", "distinct", tmp_path)
    assert res["ok"] and res["world"] == N_MULTI and res["reads"] == 24000


@need2
@pytest.mark.parametrize("name", ["paired_adapters_named", "paired", "long40_adapters"])
def test_cli_over_all_devices_is_byte_equal(name):
    argv = dict(cases.load())[name]
    devs = ",".join(str(d) for d in range(N_MULTI))
    r = subprocess.run([QUACK] + argv, capture_output=True, cwd=cases.inp(""), timeout=300,
                       env=dict(os.environ, QUACK_DEVICES=devs, QUACK_HIP_BATCH_MB="1"))
    assert r.returncode == 0, r.stderr
    assert r.stdout == cases.golden_svg(name)


@need2
def test_bench_nccl_ranks(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N_MULTI), "--reads", "1000000",
                        "--steps", "5", "--warmup", "2"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == N_MULTI and line["ranks"]["backend"] == "rccl"
