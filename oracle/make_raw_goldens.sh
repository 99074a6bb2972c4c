#!/bin/bash
# TEST INFRASTRUCTURE (build container only).  The reference's RAW counter tables —
# `bases[]` as read_fastq returns it (quack.c:223-226), before transform — for every
# case of tests/golden/cases.tsv and for BASELINE.json configs[0], taken from the
# UNMODIFIED prebuilt reference binary through the LD_PRELOAD observer oracle/ref_peek.c.
#   tests/golden/raw/<case>.<k>.u64.gz    k = 0 (forward / unpaired), 1 (reverse)
# = max_length x 97 little-endian u64 (the 776-byte base_information records), gzip -n -9.
# tests/test_oracle_pins.py compares the oracle with them, tests/test_gpu_raw_pins.py the
# HIP path (no oracle in between).
set -euo pipefail
REF=${REF:-/root/reference/bin/Linux_x86_64_kernel_3.10.0/quack}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/tests/golden
mkdir -p "$G/raw" "$ROOT/oracle/_build"
gcc -O2 -shared -fPIC -o "$ROOT/oracle/_build/ref_peek.so" "$ROOT/oracle/ref_peek.c" -ldl
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
run() {   # name, directory, argv...
  local name=$1 dir=$2; shift 2
  rm -f "$TMP"/peek.*
  set +e
  (cd "$dir" && QUACK_PEEK_OUT="$TMP/peek" LD_PRELOAD="$ROOT/oracle/_build/ref_peek.so" "$REF" "$@" > /dev/null 2> /dev/null)
  set -e
  local k
  for k in 0 1; do
    [ -f "$TMP/peek.$k" ] || continue
    gzip -n -9 -c "$TMP/peek.$k" > "$G/raw/$name.$k.u64.gz"
    echo "raw $name.$k $(stat -c %s "$TMP/peek.$k") bytes -> $(stat -c %s "$G/raw/$name.$k.u64.gz")"
  done
}
grep -v '^#' "$G/cases.tsv" | while IFS=$'\t' read -r name args; do
  [ -z "$name" ] && continue
  # shellcheck disable=SC2086
  run "$name" "$G/inputs" $args
done
if [ -x "$ROOT/tools/gen_fastq" ]; then
  "$ROOT/tools/gen_fastq" "$TMP/config1.fq.gz" 100000 150 150 12345
  [ "$(sha256sum "$TMP/config1.fq.gz" | cut -d' ' -f1)" = "$(cat "$G/svg/config1.input.sha256")" ]
  run config1 "$TMP" -u config1.fq.gz
  "$ROOT/tools/gen_fastq" "$TMP/config4_R1.fq.gz" 100000 150 150 4
  "$ROOT/tools/gen_fastq" "$TMP/config4_R2.fq.gz" 100000 150 150 5 2 30
  (cd "$TMP" && sha256sum config4_R1.fq.gz config4_R2.fq.gz) | cmp - "$G/svg/config4.input.sha256"
  run config4 "$TMP" -1 config4_R1.fq.gz -2 config4_R2.fq.gz
fi
