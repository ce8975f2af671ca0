#!/bin/bash
# diagnostic build of the library with index assertions in the device inflater's table builder (-DZR_INFLATE_BOUNDS)
set -e
cd "$(dirname "$0")/../../zlib-ng_amd/csrc"
mkdir -p ../../tools/micro/bin/obj_bounds
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -Wno-unused-result -DZR_INFLATE_BOUNDS --offload-arch=gfx950 -c $f -o ../../tools/micro/bin/obj_bounds/${f%.hip}.o &
done
for f in *.cpp; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -x c++ -c $f -o ../../tools/micro/bin/obj_bounds/${f%.cpp}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/bin/libzng_rocm_bounds.so ../../tools/micro/bin/obj_bounds/*.o
echo built tools/micro/bin/libzng_rocm_bounds.so
