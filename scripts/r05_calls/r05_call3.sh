#!/bin/bash
# r05 call 3: g256p with a FOUR-slot ring (two DMA batches behind the counted wait) + two more attribution builds of the three-slot
# loop (every DMA out of range = issued, nothing fetched; no counted wait = fragments read whether landed or not)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c3_g256p_four_slots.log; : > $OUT
cd tools/probe
for v in "-DNSLOT4" "-DNSLOT4 -DABL_M0X4" "" "-DABL_OOBDMA" "-DABL_NOWAIT" "-DNSLOT4 -DABL_NODMA"; do
  n=g256p_$(echo "$v" | sed 's/-D//g; s/ /_/g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -o /tmp/$n g256p_probe.hip 2>/dev/null
  shapes=("50432 768 3072" "50432 2304 768" "50432 768 768")
  [ "$v" = "-DNSLOT4" ] && shapes=("256 256 768" "1024 768 768" "2048 2304 768" "50432 768 3072" "50432 2304 768" "50432 3072 768" "50432 768 768" "5120 2048 8192")
  for shape in "${shapes[@]}"; do
    echo "== $n $shape" | tee -a $OUT
    timeout 60 /tmp/$n $shape 2>&1 | tail -2 | tee -a $OUT
  done
done
cd $GRAFT_REPO_ROOT; python tools/probe/g128x256_vs_production.py 2>/dev/null | tee -a $OUT
