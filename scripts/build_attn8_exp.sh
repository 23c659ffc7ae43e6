#!/bin/bash
# experiment builds of the 8-wave attention kernel next to the shipped library: scripts/libtsd_var.so (-DTSD_ATTN8_VARIANTS: every timing /
# A-B variant, picked by TSD_ATTN8_VAR) and scripts/libtsd_ts.so (-DTSD_ATTN8_TS: per-phase s_memtime ticks); only kernels_attn8.hip is
# recompiled, the other objects are the shipped build's
set -e
cd "$(dirname "$0")/../stable-diffusion.mojo_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value"
mkdir -p build_exp
hipcc $F -DTSD_ATTN8_VARIANTS $EXTRA -c kernels_attn8.hip -o build_exp/attn8_var.o &
hipcc $F -DTSD_ATTN8_TS $EXTRA -c kernels_attn8.hip -o build_exp/attn8_ts.o &
for v in $TSVARS; do hipcc $F -DTSD_ATTN8_TS -DTSD_ATTN8_DEFAULT_VAR=$v -c kernels_attn8.hip -o build_exp/attn8_ts$v.o & done
wait
OTH=$(ls build/*.o | grep -v kernels_attn8)
hipcc -shared -fPIC --offload-arch=gfx950 $OTH build_exp/attn8_var.o -o ../../scripts/libtsd_var.so -ldl
hipcc -shared -fPIC --offload-arch=gfx950 $OTH build_exp/attn8_ts.o -o ../../scripts/libtsd_ts.so -ldl
for v in $TSVARS; do hipcc -shared -fPIC --offload-arch=gfx950 $OTH build_exp/attn8_ts$v.o -o ../../scripts/libtsd_ts$v.so -ldl; done
