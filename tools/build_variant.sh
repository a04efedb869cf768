#!/bin/bash
# Build an experimental variant of the library next to the product one: tools/build_variant.sh <name> [hipcc flags...]
# -> simplyp_amd/csrc/variants/libsimplyp_<name>.so ; run with SIMPLYP_HIP_LIB=<that path> python bench.py ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/simplyp_amd/csrc/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" \
    -o $ROOT/simplyp_amd/csrc/variants/libsimplyp_$NAME.so $ROOT/simplyp_amd/csrc/simplyp_hip.hip
echo $ROOT/simplyp_amd/csrc/variants/libsimplyp_$NAME.so
