#!/bin/bash
# Diagnostic (GPU box): time conv_ppc_kernel<8> with one component removed at a time (results are garbage, only the time matters).
#   bash scripts/ppt_ablate.sh     -> gpurun_out/ppt_ablate.log
# The ablated builds are linked into a SCRATCH library (never over the shipped libmisamd.so) that the loader picks up through MISAMD_LIB; a failing step
# therefore cannot leave a wrong-result library in the tree.
set -e
cd "$(dirname "$0")/.."
CS=mdeical_image_segmentation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function -Wno-unused-variable"
F=${PPT_FILE:-conv_pp}          # the source the ablation macros live in: conv_pp (conv_ppc_kernel) or conv_ppd (conv_ppd_kernel)
OBJS=$(ls $CS/*.o | grep -v "/$F.o")
SCRATCH=$(mktemp -d /tmp/ppt_ablate.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
out=gpurun_out/ppt_ablate.log
: > $out
for m in ${PPT_MODES:-NONE PPT_NO_MFMA PPT_NO_DMA}; do
  /opt/rocm/bin/hipcc $FLAGS -D$m -c $CS/$F.hip -o $SCRATCH/conv_pp_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $SCRATCH/libmisamd_abl.so $OBJS $SCRATCH/conv_pp_abl.o -ldl
  echo "== $m" >> $out
  MISAMD_LIB=$SCRATCH/libmisamd_abl.so python scripts/bench_one_conv.py ${PPT_LAYERS:-512 64 128 256 128 128 128 256 256} >> $out 2>&1
done
cat $out
