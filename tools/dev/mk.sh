#!/bin/bash
# usage: tools/dev/mk.sh <tag> [extra hipcc flags]  -> build/lib_<tag>.so
tag=$1; shift
repo=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $repo/build
cd $repo/ltransv.2b_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "$@" -o $repo/build/lib_$tag.so lt_engine.hip 2>&1 | grep -E "error|spill" | head
ls -la $repo/build/lib_$tag.so | awk '{print $5, $9}'
