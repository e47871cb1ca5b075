#!/bin/bash
# AddressSanitizer + UBSan over the host-side C code (FASTA reader, weight tables) on the CPU:
# the golden FASTA files and 12 000 random mutations of them (bytes flipped, '>' / LF / CR
# inserted, truncation).  GPU sanitizers are not available on the pool; this covers the code
# that parses untrusted input.       bash tools/asan_host.sh
set -e
cd "$(dirname "$0")/.."
gcc -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -std=gnu11 -Iinclude -o /tmp/gkm_asan_host \
    tools/asan_host.c gkmqc_amd/csrc/gkm_host.c -lm -lpthread
for pair in "quirks_pos.fa quirks_neg.fa" "motif_pos.fa motif_neg.fa"; do
  set -- $pair
  /tmp/gkm_asan_host tests/golden/$1 tests/golden/$2
done
g++ -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -Iinclude -o /tmp/gkm_asan_bitslice \
    tools/asan_bitslice.cpp gkmqc_amd/csrc/bitslice_cpu_probe.cpp
/tmp/gkm_asan_bitslice
echo "asan/ubsan: clean"
