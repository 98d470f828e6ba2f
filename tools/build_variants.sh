#!/bin/bash
# Builds libvkr_postfx variants that differ in -D flags of one source (kernel experiments):
#   bash tools/build_variants.sh ssr.hip name1 "-DTRACE_WY=2" name2 "-DTRACE_WY=2 -DTRACE_ROUND=8" ...
# -> vk-renderer_amd/csrc/variants/<name>/libvkr_postfx.so; select with
#    V=$PWD/vk-renderer_amd/csrc/variants/<name>; LD_LIBRARY_PATH=$V VKR_POSTFX_LIB=$V/libvkr_postfx.so python bench.py
# (LD_LIBRARY_PATH so that libvkr_host.so, which finds the library through its RUNPATH, takes the same one)
set -eu
cd "$(dirname "$0")/../vk-renderer_amd/csrc"
SRC=$1; shift
make -s
mkdir -p variants
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DVKR_CONTRACT=2 -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function"
OTHERS=$(ls build/*.o | grep -v "build/${SRC%.hip}.o")
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -c $SRC -o variants/${SRC%.hip}_$name.o
  mkdir -p variants/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name/libvkr_postfx.so $OTHERS variants/${SRC%.hip}_$name.o -ldl
  echo built variants/$name/libvkr_postfx.so
done
