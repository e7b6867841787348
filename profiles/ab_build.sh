#!/bin/bash
# Builds a variant of librr_hip.so for an A/B measurement: bash profiles/ab_build.sh <name> [-D... flags]
# The variant goes to ab/librr_hip_<name>.so (untracked; it travels to the GPU box); run anything against it with
# RR_HIP_LIB=ab/librr_hip_<name>.so.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wall \
    -I "$ROOT/rust-raytracer_amd/csrc" -I "$ROOT/include" "$@" \
    -o "$ROOT/ab/librr_hip_$NAME.so" "$ROOT/rust-raytracer_amd/csrc/rr_hip.hip"
echo "built ab/librr_hip_$NAME.so"
