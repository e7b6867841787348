#!/bin/bash
# Builds a variant of librr_hip.so for an A/B measurement: bash profiles/ab_build.sh <name> [-D... flags]
# The variant goes to ab/librr_hip_<name>.so (untracked; it travels to the GPU box); run anything against it with
# RR_HIP_LIB=ab/librr_hip_<name>.so.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/ab"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -I $ROOT/rust-raytracer_amd/csrc -I $ROOT/include"
# (the library's two translation units, side by side; the second with the scheduler flag the build gives it)
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "$TMP/a.o" "$ROOT/rust-raytracer_amd/csrc/rr_hip.hip" &
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=iterative-minreg "$@" -c -o "$TMP/b.o" "$ROOT/rust-raytracer_amd/csrc/rr_paths_minreg.hip" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/ab/librr_hip_$NAME.so" "$TMP/a.o" "$TMP/b.o"
echo "built ab/librr_hip_$NAME.so"
