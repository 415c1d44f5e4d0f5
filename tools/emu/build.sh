#!/bin/bash
# tools/emu/build.sh [asan|race] — compile the UNMODIFIED kernel + host source of consul_amd/csrc for the build container's host cores against
# the wave64 lock-step emulator (tools/emu/hip/hip_runtime.h): tools/emu/_build/libswimsim_emu.so (or _asan.so under -fsanitize=address,undefined).
# TEST INFRASTRUCTURE: the library reports backend "hip-emulated", which consul_amd/lib.py refuses.
# The only textual changes, made on a scratch copy: `extern __shared__ T name[];` (dynamic LDS) becomes a pointer to the emulator's LDS
# buffer, the amdgpu_waves_per_eu attribute (which the host target does not know) is dropped, and the backend string.
set -e
cd "$(dirname "$0")"; HERE=$PWD; ROOT=$(cd ../.. && pwd)
B=$HERE/_build; S=$B/x.$$/csrc; mkdir -p "$S"; ln -sfn "$ROOT/include" "$B/include"
for f in swim_host.hip swim_kernels.hip swim_device.h; do
  sed -E -e 's/extern __shared__ ([A-Za-z0-9_]+) ([A-Za-z0-9_]+)\[\];/static \1* const \2 = (\1*)emu::dyn_lds;/' \
         -e 's/__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)//' \
         -e 's/"hip-gfx950"/"hip-emulated"/' "${EMU_SRC:-$ROOT/consul_amd/csrc}/$f" > "$S/$f"
done
printf '#include "swim_host.hip"\n#include "%s/emu_engine.inc"\n' "$HERE" > "$S/tu.cpp"
OUT=${EMU_OUT:-$B/libswimsim_emu.so}; SAN=""
EXTRA=""
if [ "$1" = race ]; then    # the kernels' loads and stores instrumented (-fsanitize=thread), tools/emu/emu_race.cpp instead of the ThreadSanitizer runtime
  OUT=${EMU_OUT:-$B/libswimsim_emu_race.so}; SAN="-fsanitize=thread -DEMU_RACE"
  ${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++} -std=c++17 -O1 -g -fPIC -c "$HERE/emu_race.cpp" -o "$B/emu_race.$$.o"; EXTRA="$B/emu_race.$$.o"
fi
if [ "$1" = asan ]; then OUT=${EMU_OUT:-$B/libswimsim_emu_asan.so}; SAN="-fsanitize=address,undefined -fno-sanitize=pointer-overflow,function -fno-sanitize-recover=undefined -fno-omit-frame-pointer"; fi
${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++} -std=c++17 -O1 -g -ffp-contract=off -fno-strict-aliasing -fwrapv -fPIC -shared $SAN $EMU_CXXFLAGS -Wno-unused-value -Wno-unknown-attributes -Wno-ignored-attributes \
    -I "$HERE" -x c++ "$S/tu.cpp" ${EXTRA:+-x none $EXTRA} -o "$OUT"
echo "$OUT"
rm -rf "$B/x.$$" "$B/emu_race.$$.o"
