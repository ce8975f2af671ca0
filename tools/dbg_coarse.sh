set -x
cd /root/repo
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import synth
open('/tmp/in.bin','wb').write(synth.silesia_like(6<<20, seed=17).tobytes())
PY
gcc -std=c11 -g -O0 -DZNG_ROCM_STANDALONE_CHECK -DROCM_MIN_BYTES=1024 -DROCM_INFLATE_MIN_BYTES=1 -DROCM_DEFLATE_BLOCK_BYTES=1048576 -Iinclude -Itests/c -Iintegration/arch/rocm tests/c/coarse_driver.c integration/arch/rocm/rocm_deflate.c integration/arch/rocm/rocm_inflate.c integration/arch/rocm/rocm_slots.c integration/arch/rocm/rocm_features.c -o /tmp/cd -Lzlib-ng_amd/lib -lzng_rocm -Wl,-rpath,$PWD/zlib-ng_amd/lib
which gdb rocgdb
(gdb -batch -ex run -ex bt --args /tmp/cd d 6 1 100000 4096 7 /tmp/in.bin /tmp/out.z 2>&1 || /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args /tmp/cd d 6 1 100000 4096 7 /tmp/in.bin /tmp/out.z 2>&1) | tail -30
