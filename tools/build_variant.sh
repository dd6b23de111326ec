#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=V ...]: a build of the product library with extra macros into tools/_ab/<name>.so (scratch)
set -e
name=$1; shift
mkdir -p tools/_ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared \
  -I include -I bnn_priors_amd/csrc "$@" bnn_priors_amd/csrc/sgmcmc_hip.hip -o tools/_ab/$name.so
echo built tools/_ab/$name.so
