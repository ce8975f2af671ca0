#!/bin/bash
# diagnostic build of the library that counts why the device inflater's hand-written loop leaves (-DZR_INFLATE_STATS) and
# reads ZNG_ROCM_INFLATE_RING (-DZR_MEASURE_FORMS: ring sizes 4096 .. 32768 for whole streams)
set -e
# EXITS=1 ./inflate_stats.sh adds the per-reason exit counters (an atomic per exit: times are no longer meaningful)
EXTRA=${EXITS:+-DZR_INFLATE_STATS_EXITS}
# NOSTATS=1 ./inflate_stats.sh: the measurement forms alone (ZNG_ROCM_INFLATE_RING, ZNG_ROCM_PART_RING), no counters or stamps
STATS=-DZR_INFLATE_STATS
if [ -n "$NOSTATS" ]; then STATS=; fi
cd "$(dirname "$0")/../../zlib-ng_amd/csrc"
mkdir -p ../../tools/micro/bin/obj_stats
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -Wno-unused-result $STATS -DZR_MEASURE_FORMS $EXTRA --offload-arch=gfx950 -c $f -o ../../tools/micro/bin/obj_stats/${f%.hip}.o &
done
for f in *.cpp; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -x c++ -c $f -o ../../tools/micro/bin/obj_stats/${f%.cpp}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/bin/libzng_rocm_stats.so ../../tools/micro/bin/obj_stats/*.o
echo built tools/micro/bin/libzng_rocm_stats.so
