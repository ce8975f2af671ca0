#!/bin/bash
# diagnostic build of the library with phase time stamps in lz_rows_kernel (-DZR_ROWS_STAMPS) -> tools/micro/bin/
set -e
cd "$(dirname "$0")/../../zlib-ng_amd/csrc"
mkdir -p ../../tools/micro/bin/obj_stamps
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -Wno-unused-result -DZR_ROWS_STAMPS --offload-arch=gfx950 -c $f -o ../../tools/micro/bin/obj_stamps/${f%.hip}.o &
done
for f in *.cpp; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -x c++ -c $f -o ../../tools/micro/bin/obj_stamps/${f%.cpp}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/bin/libzng_rocm_stamps.so ../../tools/micro/bin/obj_stamps/*.o
echo built tools/micro/bin/libzng_rocm_stamps.so
