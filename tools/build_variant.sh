#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=V ...]: a build of the product library with extra macros into tools/_ab/<name>.so (scratch)
set -e
name=$1; shift
mkdir -p tools/_ab
# stamped "<source sha>+<flags>": the loader accepts it for this tree and measurements record which variant ran
SHA=$(python -c "from bnn_priors_amd import _hip; print(_hip.source_sha())")
FLAGS=$(echo "$*" | tr -d ' "')
hipcc -DSGMCMC_SOURCE_SHA="\"$SHA+$FLAGS\"" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared \
  -I include -I bnn_priors_amd/csrc "$@" bnn_priors_amd/csrc/sgmcmc_hip.hip -o tools/_ab/$name.so
echo built tools/_ab/$name.so
