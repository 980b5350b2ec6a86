#!/bin/bash
# build_ab.sh NAME "-DFLAG ..." [file.hip]: an A/B build of libmi355conv.so with extra flags on ONE translation unit
# (default conv_igemm.hip), written to ab/NAME.so (git-ignored, travels to the GPU box; select it with MI355_LIB)
set -e
name="$1"; flags="$2"; tu="${3:-conv_igemm.hip}"
cd "$(dirname "$0")/../medical-image-segmentation-and-classification_amd/csrc"
make -s -j4 >/dev/null
obj=build/ab_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable $flags -c $tu -o $obj
mkdir -p ../../ab
others=$(ls build/*.o | grep -v "build/ab_\|build/drain_\|build/probe.o\|build/${tu%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/${name}.so $obj $others -lz -lpthread
echo "ab/${name}.so"
