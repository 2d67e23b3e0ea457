#!/bin/bash
# A build VARIANT of libvgpa_hip.so beside the product library, made in this (CPU-only) container so that a GPU call only runs it:
#   tools/build_variant.sh <name> "<extra compiler flags>" [sources to recompile, default: ode_mfma_m3.hip]
# -> vgpa_amd/lib/variants/libvgpa_hip_<name>.so (objects of the other sources are taken from the product build, which must be current).
# Use: VGPA_LIB=$PWD/vgpa_amd/lib/variants/libvgpa_hip_<name>.so python bench.py ...
set -e
name=$1; flags=$2; shift 2 || true
srcs=${*:-ode_mfma_m3.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/vgpa_amd/lib/variants; obj=$root/vgpa_amd/build/variant_$name
mkdir -p $out $obj
objs=""
for o in $root/vgpa_amd/build/*.o; do
  b=$(basename $o .o); hit=0
  for s in $srcs; do [ "${s%.*}" = "$b" ] && hit=1; done
  if [ $hit = 1 ]; then
    src=$root/vgpa_amd/csrc/$b.hip; [ -f $src ] || src=$root/vgpa_amd/csrc/$b.cpp
    ff=""; case $b in large_d_stage|large_d_energy) ff="-mllvm -amdgpu-mfma-vgpr-form";; esac      # per-file flags of vgpa_amd/build.py (FILE_CFLAGS)
    /opt/rocm/bin/hipcc -x hip -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $ff $flags -c $src -o $obj/$b.o
    objs="$objs $obj/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libvgpa_hip_$name.so $objs -ldl
echo $out/libvgpa_hip_$name.so
