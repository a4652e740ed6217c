#!/bin/bash
# Diagnostic (GPU box): the stamp table of conv_ppc2_kernel (MIS_CONV_PPC2=1), as scripts/ppc_stamps.sh does for conv_ppc_kernel -> gpurun_out/ppc2_stamps.txt
set -e
cd "$(dirname "$0")/.."
CS=mdeical_image_segmentation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=$(ls $CS/*.o | grep -v "/conv_ppc2.o")
SCRATCH=$(mktemp -d /tmp/ppc2_stamps.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
/opt/rocm/bin/hipcc $FLAGS -DMIS_PP_STAMPS -c $CS/conv_ppc2.hip -o $SCRATCH/conv_ppc2_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $SCRATCH/libmisamd_st.so $OBJS $SCRATCH/conv_ppc2_st.o -ldl
out=gpurun_out/ppc2_stamps.txt
: > $out
for L in "64 512 512" "256 128 128" "512 64 128"; do
  MIS_CONV_PPC2=1 STAMP_SYMBOL=mis_debug_ppc2_stamps MISAMD_LIB=$SCRATCH/libmisamd_st.so python scripts/pp_stamps.py $L >> $out 2>&1
done
cat $out
