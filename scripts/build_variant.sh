#!/bin/bash
# build_variant.sh <name> <source.hip> "<extra flags>": libhpslice_<name>.so = the shipped objects with ONE source
# recompiled under extra flags (A/B runs: HPS_LIB=hipace_amd/csrc/libhpslice_<name>.so python bench.py ...)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../hipace_amd/csrc"
make -s libhpslice.so
extra=""
[ "$src" = particles_tiled.hip ] && extra="-mllvm -disable-lsr"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $extra $flags -c $src -o ${src%.hip}.$name.vo
objs=""
for s in particles.hip particles_tiled.hip sort.hip poisson.hip multigrid.hip multigrid2.hip engine.hip beam.hip laser.hip ring.hip ionization.hip; do
  if [ "$s" = "$src" ]; then objs="$objs ${s%.hip}.$name.vo"; else objs="$objs ${s%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libhpslice_$name.so $objs -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib
echo built libhpslice_$name.so
