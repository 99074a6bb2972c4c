"""CPU-only: AddressSanitizer + UBSan build of the host tokenizer / gzip decoder,
fuzzed against the oracle tokenizer (tests/c/host_fuzz.c).  (GPU ASan is not
available on the pool; this is the sanitizer tier for the C host code.)"""
import os
import subprocess

import cases

SRC = [os.path.join(cases.ROOT, p) for p in (
    "tests/c/host_fuzz.c", "quack_amd/host/reader.c", "quack_amd/host/source.c", "quack_amd/host/inflate_fast.c", "quack_amd/host/crc32_fold.c", "quack_amd/host/pinflate.c", "oracle/quack_oracle.c")]


def test_tokenizer_and_inflate_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    subprocess.check_call(
        ["gcc", "-O1", "-g", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-pthread",
         "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
         "-I" + os.path.join(cases.ROOT, "include"), "-I" + os.path.join(cases.ROOT, "quack_amd", "host"),
         "-I" + os.path.join(cases.ROOT, "oracle"), "-o", exe] + SRC + ["-lz"])
    r = subprocess.run([exe, "250", "11"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failures" in r.stdout


def test_reader_threads_under_tsan(tmp_path):
    """ThreadSanitizer over the same fuzz: producer threads of source.c (BGZF pool,
    multi-threaded gzip of pinflate.c) against the consuming tokenizer"""
    exe = str(tmp_path / "host_fuzz_tsan")
    subprocess.check_call(
        ["gcc", "-O1", "-g", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-pthread",
         "-fsanitize=thread", "-fno-omit-frame-pointer",
         "-I" + os.path.join(cases.ROOT, "include"), "-I" + os.path.join(cases.ROOT, "quack_amd", "host"),
         "-I" + os.path.join(cases.ROOT, "oracle"), "-o", exe] + SRC + ["-lz"])
    r = subprocess.run([exe, "80", "5"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "0 failures" in r.stdout and "ThreadSanitizer" not in r.stderr
