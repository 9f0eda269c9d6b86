#!/bin/bash
# Development aid: build remhos_amd/librmh_<name>.so from the current sources with extra compiler flags, so that
# several kernel variants can be timed in ONE gpurun call (tools/kbench.py).  Never shipped: *.so is git-ignored.
#   bash tools/build_variant.sh <name> [-DFLAG ...]
set -eu
name=$1; shift
cd "$(dirname "$0")/../remhos_amd/csrc"
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-honor-nans "$@" -c rmh_api.hip -o build/rmh_api_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC build/rmh_api_$name.o build/rmh_driver.o build/rmh_host.o build/rmh_case_api.o -o ../librmh_$name.so -pthread -ldl
rm -f build/rmh_api_$name.o   # (2 MB each: they would travel with every gpurun snapshot)
echo built remhos_amd/librmh_$name.so
