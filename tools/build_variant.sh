#!/bin/bash
# build_variant.sh NAME [ENV=VALUE ...] -- an A/B build of libnbx.so with other generator settings (tools/gen_sgpr_loop.py reads its
# knobs from the environment): copies csrc/, regenerates the loop include there, links tools/ab/NAME/libnbx.so.  Run with
# NBX_LIB=tools/ab/NAME/libnbx.so.  Development harness, not part of the product.
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$here/..
name=$1; shift
dir=$here/ab/$name
rm -rf "$dir"; mkdir -p "$dir/pkg/csrc" "$dir/include"
cp "$root"/nbody-demo-2023_amd/csrc/* "$dir/pkg/csrc/"
cp "$root"/include/nbx.h "$dir/include/"
env "$@" python3 "$here/gen_sgpr_loop.py" "$dir/pkg/csrc/nbx_sgpr_loop.inc"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt"
# csrc includes "../../include/nbx.h" relative to pkg/csrc -> $dir/include
(cd "$dir/pkg/csrc" && hipcc $flags -c nbx_api.hip -o ../nbx_api.o && hipcc $flags -c nbx_group.hip -o ../nbx_group.o &&
 hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c nbx_ic.cpp -o ../nbx_ic.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$dir/libnbx.so" "$dir"/pkg/*.o -ldl
echo "built $dir/libnbx.so"
