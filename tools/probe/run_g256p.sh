#!/bin/bash
# First GPU contact of tools/probe/g256p_probe.hip (written in round 4, never run): every run under `timeout` (an untested LDS-DMA ring
# can hang), the exposed-epilogue variant (park = 0: ring + main loop + way out) before the parked one, the smallest problem first, and the
# production kernel on the same shapes beside it.  Usage (from the repo root, on the GPU box): bash tools/probe/run_g256p.sh [outfile]
set -u
OUT=${1:-gpurun_out/g256p_first_contact.log}; mkdir -p "$(dirname "$OUT")"; : > "$OUT"
P=tools/probe/g256p_probe
[ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $P tools/probe/g256p_probe.hip 2>/dev/null
run() { echo "== $*" | tee -a "$OUT"; timeout 40 $P "$@" 2>&1 | tail -3 | tee -a "$OUT"; echo "rc=${PIPESTATUS[0]}" | tee -a "$OUT"; }
run 256 256 768 0        # ONE tile, one workgroup: ring, fragments, MFMA order, permlane pieces
run 256 256 768 1
run 1024 768 768 0       # 12 tiles on 12 workgroups
run 2048 2304 768        # 72 tiles: still one tile per workgroup
run 50432 768 768        # 591 tiles on 256 workgroups: the walk across tile boundaries (2.31 tiles each)
run 50432 2304 768
run 50432 3072 768
run 50432 768 3072
python tools/probe/g128x256_vs_production.py 2>/dev/null | tee -a "$OUT"
