#!/bin/bash
# usage: build/mk.sh <tag> [extra hipcc flags]  -> build/lib_<tag>.so
tag=$1; shift
cd /root/repo/ltransv.2b_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "$@" -o /root/repo/build/lib_$tag.so lt_engine.hip 2>&1 | grep -E "error|spill" | head
ls -la /root/repo/build/lib_$tag.so | awk '{print $5, $9}'
