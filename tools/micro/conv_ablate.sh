#!/bin/bash
# Where does the fixed cost of a CONV_TAPS launch go?  Builds variants of the library with parts of the conv kernel
# disabled (HP_ABL bits, conv_mfma.hip) into gpurun_out/abl/ and prints the build lines; run tools/micro/conv_sweep.py
# with HIPPIE_HIP_LIB=<variant> on the GPU box.
set -e
cd "$(dirname "$0")/../../hippie_amd/csrc"
make -s
mkdir -p ../../tools/micro/variants
for v in ${VARIANTS:-1 2 3 4 8}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DHP_ABL=$v -c conv_mfma.hip -o /tmp/conv_abl$v.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 program.o model.o /tmp/conv_abl$v.o ops_small.o -o ../../tools/micro/variants/libhippie_abl$v.so
  echo "built variant $v"
done
