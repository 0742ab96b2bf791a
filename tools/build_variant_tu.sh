#!/bin/bash
# Variant of libeae.so that differs from the product library in ONE translation unit (seconds instead of a minute):
#   tools/build_variant_tu.sh <tag> <tu without .hip> <flags...>   -> <pkg>/libeae_<tag>.so   (the other objects come from csrc/_obj)
set -e
TAG=$1; TU=$2; shift 2
PKG=$(cd "$(dirname "$0")/../hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd" && pwd)
OBJ=$PKG/csrc/_obj_$TAG
mkdir -p "$OBJ"
cp "$PKG"/csrc/_obj/*.o "$OBJ"/
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c "$PKG/csrc/$TU.hip" -o "$OBJ/$TU.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/libeae_$TAG.so" "$OBJ"/*.o
echo "$PKG/libeae_$TAG.so"
