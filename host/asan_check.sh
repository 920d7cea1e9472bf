#!/bin/bash
# Runs the C++ host's scenarios (frame streaming, save/resume, collisions in both modes, the reference-arithmetic
# mode, mixed precision) under host-side AddressSanitizer.  Build first: make -C host asan.  Needs a GPU.
set -u
export ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:halt_on_error=1
B=$(cd "$(dirname "$0")" && pwd)/_asan/nbody_asan
W=$(mktemp -d); cd "$W" || exit 1
run() {
    echo "== $*"
    timeout -k 10 120 "$B" "$@" > out.txt 2>&1; rc=$?
    grep -n "AddressSanitizer\|ERROR" out.txt | head -5
    echo "rc=$rc"
    [ $rc -eq 0 ] || { tail -20 out.txt; exit 1; }
}
run -n 500 -y 0.004 -nocollision -save -seed 5 -tree
run -n 500 -y 0.001 -nocollision -state 0000000036.data -frames-dir more -tree
run -n 3000 -y 0.003 -tree -seed 3 -save
run -y 0.002
run -n 400 -y 0.002 -seed 2
run -n 300 -y 0.002 -tree -nocollision -asgo -precision f64
run -n 2000 -y 0.002 -tree -precision mixed -norender
run -n 2000 -y 0.002 -tree -perwave
echo ALL-OK
