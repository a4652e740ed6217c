#!/bin/bash
# Diagnostic (GPU box): time wgrad_pp_row_kernel with one component removed at a time (results are garbage, only the time matters).
#   bash scripts/wgrad_ablate.sh     -> gpurun_out/wgrad_ablate.log
# The ablated builds are linked into a SCRATCH library that the loader picks up through MISAMD_LIB (never over the shipped libmisamd.so).
set -e
cd "$(dirname "$0")/.."
CS=mdeical_image_segmentation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=$(ls $CS/*.o | grep -v wgrad_pp.o)
SCRATCH=$(mktemp -d /tmp/wgrad_ablate.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
out=gpurun_out/wgrad_ablate.log
: > $out
for m in ${WPT_MODES:-NONE WPT_NO_MFMA WPT_NO_DMA}; do
  /opt/rocm/bin/hipcc $FLAGS -D$m -c $CS/wgrad_pp.hip -o $SCRATCH/wgrad_pp_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $SCRATCH/libmisamd_abl.so $OBJS $SCRATCH/wgrad_pp_abl.o -ldl
  echo "== $m" >> $out
  MISAMD_LIB=$SCRATCH/libmisamd_abl.so python scripts/bench_wgrad_layers.py 2>&1 | grep -v amdgpu.ids | cut -c1-72 >> $out
done
cat $out
