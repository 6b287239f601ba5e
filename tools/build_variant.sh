#!/bin/bash
# Build a diagnostic variant of the library: tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]
# -> mvtracker_amd/lib/libmvtracker_hip_<name>.so (the named source recompiled with the flags, every other object reused).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; src="$2"; shift 2
lib="$ROOT/mvtracker_amd/lib"
base="$(basename "$src" .hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c "$ROOT/mvtracker_amd/csrc/$base.hip" -o "$lib/${base}_$name.o"
objs=""
for o in "$lib"/*.o; do
  b="$(basename "$o" .o)"
  case "$b" in *_*_*|*_"$name") ;; esac
  if [ "$b" = "$base" ]; then continue; fi
  if [[ "$b" == *_"$name" && "$b" != "${base}_$name" ]]; then continue; fi
  # skip other variants' objects (name_variant.o): keep only the plain objects and this variant's object
  if [[ "$b" != "${base}_$name" && ! -f "$ROOT/mvtracker_amd/csrc/$b.hip" ]]; then continue; fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$lib/libmvtracker_hip_$name.so" $objs
echo "$lib/libmvtracker_hip_$name.so"
