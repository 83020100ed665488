#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA compiler flags"  ->  scratch/ab/lib_NAME.so
# Out-of-tree build of libmomcore.so (the in-tree objects stay untouched): same-box A/B runs through MOM_LIBRARY.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
base=$root/scratch/build/$1
mkdir -p "$base/x/csrc" "$base/include" "$root/scratch/ab"
cp -u "$root"/include/*.h "$base/include/"
for f in "$root"/radiativetransfer.jl_amd/csrc/*; do case "$f" in *.o) ;; *) cp -u "$f" "$base/x/csrc/";; esac; done
make -C "$base/x/csrc" -j8 EXTRA="$2" > "$base/build.log" 2>&1 || { tail -20 "$base/build.log"; exit 1; }
cp "$base/x/libmomcore.so" "$root/scratch/ab/lib_$1.so"
echo "built scratch/ab/lib_$1.so"
