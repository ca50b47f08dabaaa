#!/bin/bash
# build_variant.sh NAME [-DFLAG ...]: a second build of the library under build/NAME.so (timing experiments, same-box A/B through SBM_LIB)
set -e
cd "$(dirname "$0")/../score_based_multimodal_autoencoder_amd/csrc"
name=$1; shift
mkdir -p ../../build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o ../../build/$name.so plan.cpp kernels.hip chain.hip rowrun.hip pixrun.hip sampler.hip guidance.hip fid.hip api.hip
echo built build/$name.so
