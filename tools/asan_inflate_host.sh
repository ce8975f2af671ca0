#!/bin/sh
# Sanitizer run of the host inflate stage (CPU build only -- GPU sanitizers are not available on the pool).
# Builds the host-only translation units (inflate_host.cpp: decoder + block finder; inflate_threads.cpp: the
# multi-threaded decode of one stream) with ASan + UBSan and drives them with the fuzz generator of
# tests/test_inflate_host.py::test_fuzz_against_cpython_zlib (standalone copy: ctypes, no torch), then with larger
# streams -- intact, bit-flipped, truncated -- through the threaded decoder.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
g++ -O1 -g -shared -fPIC -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -pthread -I "$ROOT/include" "$ROOT/zlib-ng_amd/csrc/inflate_host.cpp" "$ROOT/zlib-ng_amd/csrc/inflate_threads.cpp" \
    -o /tmp/libinf_asan.so
for seed in ${SEEDS:-1 2 3}; do
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
        python3 "$ROOT/tools/fuzz_inflate_host.py" /tmp/libinf_asan.so "$seed" "${CASES:-3000}"
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
        python3 "$ROOT/tools/fuzz_inflate_host.py" /tmp/libinf_asan.so "$seed" "${BIG_CASES:-40}" threads
done
