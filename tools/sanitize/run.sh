#!/bin/bash
# The host-side decimal arithmetic (csrc/phx_dec.c) under AddressSanitizer + UBSan: 420 000 random operations (operands of up to 60
# digits, precisions 1..60, ln / exp / ** with real and integer exponents, repr(float), Decimal(float), start weights).
# Runs on the CPU:  tools/sanitize/run.sh   -> "ok 420000 calls" and no sanitizer report.
set -e
cd "$(dirname "$0")/../.."
gcc -std=gnu11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iphanotate_amd/csrc -Iinclude tools/sanitize/dec_asan.c phanotate_amd/csrc/phx_dec.c -lm -o /tmp/phx_dec_asan
UBSAN_OPTIONS=print_stacktrace=1 /tmp/phx_dec_asan
