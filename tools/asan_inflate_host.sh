#!/bin/sh
# Sanitizer run of the host inflate stage (CPU build only -- GPU sanitizers are not available on the pool).
# Builds zlib-ng_amd/csrc/inflate_host.cpp alone with ASan + UBSan and drives it with the fuzz generator of
# tests/test_inflate_host.py::test_fuzz_against_cpython_zlib (standalone copy below: ctypes, no torch).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
g++ -O1 -g -shared -fPIC -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -I "$ROOT/include" "$ROOT/zlib-ng_amd/csrc/inflate_host.cpp" -o /tmp/libinf_asan.so
for seed in ${SEEDS:-1 2 3}; do
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
        python3 "$ROOT/tools/fuzz_inflate_host.py" /tmp/libinf_asan.so "$seed" "${CASES:-3000}"
done
