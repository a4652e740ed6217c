#!/bin/bash
# Diagnostic (GPU box): per-segment cycle table of conv_ppc_kernel<8, 4> (the dominant 2-D key) from a -DMIS_PP_STAMPS build linked to a SCRATCH library (never the shipped one):
#   bash scripts/ppc_stamps.sh   -> gpurun_out/ppc_stamps.txt
# Three layer classes: deep (512 -> 512 at 64^2), 128 channels at 256^2, short-K (64 -> 128 at 512^2).  The stamps themselves cost ~10 % (s_memtime per point).
set -e
cd "$(dirname "$0")/.."
CS=mdeical_image_segmentation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=$(ls $CS/*.o | grep -v "/conv_pp.o")
SCRATCH=$(mktemp -d /tmp/ppc_stamps.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
/opt/rocm/bin/hipcc $FLAGS -DMIS_PP_STAMPS -c $CS/conv_pp.hip -o $SCRATCH/conv_pp_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $SCRATCH/libmisamd_st.so $OBJS $SCRATCH/conv_pp_st.o -ldl
out=gpurun_out/ppc_stamps.txt
: > $out
for L in "64 512 512" "256 128 128" "512 64 128"; do
  MISAMD_LIB=$SCRATCH/libmisamd_st.so python scripts/pp_stamps.py $L >> $out 2>&1
done
echo "--- the same layers on the shipped library (no stamps): ms per launch" >> $out
python scripts/bench_one_conv.py 64 512 512 256 128 128 512 64 128 >> $out 2>&1
cat $out
