#!/bin/bash
# development: variants of libgf_step.so that differ in the -D flags gf_post.hip is compiled with, for A/B runs on one box
#   tools/ab_build.sh <name> [-DFLAG ...]   ->  tools/_ab/<name>/libgf_step.so   (the other objects are the product build's)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$root/tools/_ab/$name"
cd "$root/genesis-forge_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden "$@" -c gf_post.hip -o "$root/tools/_ab/$name/gf_post.o"
objs=$(ls *.o | grep -v '^gf_post.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_ab/$name/libgf_step.so" $objs "$root/tools/_ab/$name/gf_post.o"
echo "built tools/_ab/$name/libgf_step.so"
