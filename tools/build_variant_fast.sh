#!/bin/bash
# usage: tools/build_variant_fast.sh NAME "EXTRA compiler flags" obj1.o [obj2.o ...]  ->  scratch/ab/lib_NAME.so
# Like build_variant.sh, but only the named objects are rebuilt with the EXTRA flags; every other object is taken from the
# in-tree build as it stands (copied with its timestamp).  For A/B runs of ONE kernel image (seconds instead of minutes).
set -e
name=$1; extra=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
base=$root/scratch/build/$name
mkdir -p "$base/x/csrc" "$base/include" "$root/scratch/ab"
cp -p "$root"/include/*.h "$base/include/"
cp -p "$root"/radiativetransfer.jl_amd/csrc/* "$base/x/csrc/" 2>/dev/null || true
for o in "$@"; do rm -f "$base/x/csrc/$o"; done
make -C "$base/x/csrc" -j8 EXTRA="$extra" > "$base/build.log" 2>&1 || { tail -20 "$base/build.log"; exit 1; }
grep -c hipcc "$base/build.log" | xargs echo "compiler invocations:"
cp "$base/x/libmomcore.so" "$root/scratch/ab/lib_$name.so"
echo "built scratch/ab/lib_$name.so"
