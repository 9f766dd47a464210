#!/bin/bash
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude $@ -o scratch/ldlt/ldlt_bench scratch/ldlt/ldlt_bench.hip orb-slam2-dualcam_amd/csrc/common.cpp orb-slam2-dualcam_amd/csrc/config.cpp 2>&1 | grep -E "error|warning: v" | head
ls -la scratch/ldlt/ldlt_bench
