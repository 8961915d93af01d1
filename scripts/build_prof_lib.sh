#!/bin/bash
# como_amd/lib_prof/libcomo_hip.so: the product library with csrc/track.hip compiled -DCOMO_TL_PROFILE (in-kernel phase stamps of
# the tracking level kernels: scripts/track_stamps.py, scripts/track_one_stamps.py; COMO_HIP_LIB=$PWD/como_amd/lib_prof/libcomo_hip.so).
# Needs an up-to-date como_amd/lib (python -m como_amd.build); git-ignored, travels with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
python -m como_amd.build > /dev/null
mkdir -p como_amd/lib_prof
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DCOMO_TL_PROFILE -c como_amd/csrc/track.hip -o como_amd/lib_prof/track.o
$HIPCC --offload-arch=gfx950 -shared -fPIC $(ls como_amd/lib/*.o | grep -v /track.o) como_amd/lib_prof/track.o -o como_amd/lib_prof/libcomo_hip.so
echo como_amd/lib_prof/libcomo_hip.so
