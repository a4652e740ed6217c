#!/bin/bash
# Diagnostic (GPU box): time conv_ppc_kernel<8> with one component removed at a time (results are garbage, only the time matters).
#   bash scripts/ppt_ablate.sh     -> gpurun_out/ppt_ablate.log
set -e
cd "$(dirname "$0")/.."
CS=mdeical_image_segmentation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=$(ls $CS/*.o | grep -v conv_pp.o)
out=gpurun_out/ppt_ablate.log
: > $out
for m in NONE PPT_NO_MFMA PPT_NO_DMA; do
  /opt/rocm/bin/hipcc $FLAGS -D$m -c $CS/conv_pp.hip -o /tmp/conv_pp_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mdeical_image_segmentation_amd/libmisamd.so $OBJS /tmp/conv_pp_abl.o -ldl
  echo "== $m" >> $out
  python scripts/bench_one_conv.py 64 512 512 128 256 256 256 128 128 >> $out 2>&1
done
/opt/rocm/bin/hipcc $FLAGS -c $CS/conv_pp.hip -o $CS/conv_pp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mdeical_image_segmentation_amd/libmisamd.so $OBJS $CS/conv_pp.o -ldl
cat $out
