#!/bin/bash
# TEST INFRASTRUCTURE.  Produce the golden outputs under tests/golden/ by
# running the REFERENCE ITSELF: its prebuilt binary
#   /root/reference/bin/Linux_x86_64_kernel_3.10.0/quack   (quack 1.1.1, real klib)
# quack.c cannot be compiled here (klib/kseq.h absent), so this binary is the
# only executable form of the reference.  Only runs in the build container;
# the outputs are committed (SVGs gzipped with -n for reproducible bytes).
set -euo pipefail
REF=${REF:-/root/reference/bin/Linux_x86_64_kernel_3.10.0/quack}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/tests/golden
mkdir -p "$G/svg" "$G/cli"
cd "$G/inputs"
grep -v '^#' "$G/cases.tsv" | while IFS=$'\t' read -r name args; do
  [ -z "$name" ] && continue
  set +e
  # shellcheck disable=SC2086
  "$REF" $args > "$G/svg/$name.svg" 2> "$G/svg/$name.err"
  rc=$?
  set -e
  echo "$rc" > "$G/svg/$name.rc"
  gzip -n -9 -f "$G/svg/$name.svg"
  echo "golden $name rc=$rc $(stat -c %s "$G/svg/$name.svg.gz") bytes"
done
grep -v '^#' "$G/cli_cases.tsv" | while IFS=$'\t' read -r name args; do
  [ -z "$name" ] && continue
  set +e
  # shellcheck disable=SC2086
  "$REF" $args > "$G/cli/$name.out" 2> "$G/cli/$name.err"
  echo $? > "$G/cli/$name.rc"
  set -e
done
echo "cli goldens: $(ls "$G/cli" | wc -l) files"

# BASELINE.json configs[0]: `quack -u` on 100k-read synthetic 150 bp gzipped
# FASTQ.  The 16.8 MB input is not committed; tools/gen_fastq regenerates it
# bit-identically (splitmix64, seed 12345) wherever the test runs.
if [ -x "$ROOT/tools/gen_fastq" ]; then
  TMP=$(mktemp -d)
  "$ROOT/tools/gen_fastq" "$TMP/config1.fq.gz" 100000 150 150 12345
  (cd "$TMP" && "$REF" -u config1.fq.gz > "$G/svg/config1.svg" 2> "$G/svg/config1.err"; echo $? > "$G/svg/config1.rc")
  sha256sum "$TMP/config1.fq.gz" | cut -d' ' -f1 > "$G/svg/config1.input.sha256"
  gzip -n -9 -f "$G/svg/config1.svg"
  echo "golden config1 $(stat -c %s "$G/svg/config1.svg.gz") bytes"
  # BASELINE.json configs[3]'s shape, CPU-runnable: paired, 100k reads per mate, 150 bp, R2 qualities skewed
  # lower (Q in [2,30]), seeds 4 / 5 — two independent accumulations and the mirrored second panel
  # (quack.c:879,911-921).  Inputs regenerated bit-identically by tools/gen_fastq wherever the test runs.
  "$ROOT/tools/gen_fastq" "$TMP/config4_R1.fq.gz" 100000 150 150 4
  "$ROOT/tools/gen_fastq" "$TMP/config4_R2.fq.gz" 100000 150 150 5 2 30
  (cd "$TMP" && "$REF" -1 config4_R1.fq.gz -2 config4_R2.fq.gz > "$G/svg/config4.svg" 2> "$G/svg/config4.err"; echo $? > "$G/svg/config4.rc")
  (cd "$TMP" && sha256sum config4_R1.fq.gz config4_R2.fq.gz) > "$G/svg/config4.input.sha256"
  gzip -n -9 -f "$G/svg/config4.svg"
  echo "golden config4 $(stat -c %s "$G/svg/config4.svg.gz") bytes"
  rm -rf "$TMP"
fi
