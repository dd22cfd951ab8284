#!/bin/bash
# development: variants of libgf_step.so that differ in the -D flags ONE source file is compiled with, for A/B runs on one box
#   tools/ab_build.sh <name> <file.hip> [-DFLAG ...]   ->  tools/_ab/<name>/libgf_step.so   (the other objects are the product build's)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
obj=${src%.hip}.o
mkdir -p "$root/tools/_ab/$name"
cd "$root/genesis-forge_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden "$@" -c "$src" -o "$root/tools/_ab/$name/$obj"
objs=$(ls *.o | grep -v "^$obj\$")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_ab/$name/libgf_step.so" $objs "$root/tools/_ab/$name/$obj"
echo "built tools/_ab/$name/libgf_step.so"
