#!/bin/bash
# Diagnostic (GPU box): bash scripts/mfma_peak.sh -> gpurun_out/mfma_peak.txt   (see scripts/mfma_peak.hip)
set -e
cd "$(dirname "$0")/.."
SCRATCH=$(mktemp -d /tmp/mfmapeak.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o $SCRATCH/mfma_peak scripts/mfma_peak.hip
mkdir -p gpurun_out
timeout -k 10 120 $SCRATCH/mfma_peak 2>&1 | tee gpurun_out/mfma_peak.txt
