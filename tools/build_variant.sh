#!/bin/bash
# usage: tools/build_variant.sh <out.so> [extra hipcc flags...] — the product library's exact build command with another output path (A/B builds under _ab/)
out=$1; shift
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-value -shared -fPIC "$@" -o "$out" consul_amd/csrc/swim_host.hip
