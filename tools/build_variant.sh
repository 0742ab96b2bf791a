#!/bin/bash
# Build a diagnostic VARIANT of libeae.so beside the product library (which stays untouched):
#   tools/build_variant.sh stamps -DEAE_STAMPS        -> <pkg>/libeae_stamps.so   (use with EAE_LIB_PATH=<that file>)
set -e
TAG=$1; shift
PKG=$(dirname "$0")/../hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd
PKG=$(cd "$PKG" && pwd)
OBJ=$PKG/csrc/_obj_$TAG
mkdir -p "$OBJ"
pids=()
for s in eae_api eae_conv_launch eae_edge_launch eae_wgrad_launch eae_fc_launch eae_misc eae_head eae_mlp; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c "$PKG/csrc/$s.hip" -o "$OBJ/$s.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/libeae_$TAG.so" "$OBJ"/*.o
echo "$PKG/libeae_$TAG.so"
