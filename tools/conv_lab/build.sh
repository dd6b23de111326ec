#!/bin/bash
# builds the standalone convolution lab: `lab` (timing; no stamps) and `lab_st` (wall-clock stamps inside the kernels:
# the stamps themselves cost waits, so never time with them).  Both git-ignored; they travel with gpurun.
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I ../../include -I ../../bnn_priors_amd/csrc"
hipcc $F -DNO_STAMPS lab.hip -o lab && hipcc $F lab.hip -o lab_st
hipcc $F -DNO_STAMPS lab50.hip -o lab50 && hipcc $F lab50.hip -o lab50_st
hipcc $F atom.hip -o atom
hipcc $F lab_uni.hip -o lab_uni && hipcc $F -DUNI_STAMPS lab_uni.hip -o lab_uni_st
hipcc $F lab_pair.hip -o lab_pair
hipcc $F lab_side.hip -o lab_side
