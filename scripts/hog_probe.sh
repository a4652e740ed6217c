#!/bin/bash
# Diagnostic (GPU box): bash scripts/hog_probe.sh -> gpurun_out/hog_probe.txt   (see scripts/hog_probe.py)
set -e
cd "$(dirname "$0")/.."
SCRATCH=$(mktemp -d /tmp/hog.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o $SCRATCH/libcuhog.so scripts/cu_hog.hip
mkdir -p gpurun_out
CU_HOG_LIB=$SCRATCH/libcuhog.so timeout -k 10 300 python3 scripts/hog_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/hog_probe.txt
