#!/bin/bash
# usage: tools/dev/mk.sh <tag> [extra hipcc flags]  -> build/lib_<tag>.so
# (the two translation units of the library as __graft_entry__.build_hip compiles them; RK_FLAGS replaces the
# options that only lt_rk_kernel.hip gets)
tag=$1; shift
repo=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $repo/build/obj
cd $repo/ltransv.2b_amd/csrc
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17"
RK_FLAGS=${RK_FLAGS--mllvm -disable-machine-licm}
( /opt/rocm/bin/hipcc $F "$@" -c -o $repo/build/obj/${tag}_engine.o lt_engine.hip 2>&1 | grep -E "error|spill" | head ) &
( /opt/rocm/bin/hipcc $F $RK_FLAGS "$@" -c -o $repo/build/obj/${tag}_rk.o lt_rk_kernel.hip 2>&1 | grep -E "error|spill" | head ) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $repo/build/lib_$tag.so $repo/build/obj/${tag}_engine.o $repo/build/obj/${tag}_rk.o
rm -f $repo/build/obj/${tag}_engine.o $repo/build/obj/${tag}_rk.o
ls -la $repo/build/lib_$tag.so | awk '{print $5, $9}'
