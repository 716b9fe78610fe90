#!/bin/bash
# r05 call 13: row-band planner knobs re-checked on the current kernel (same box, one run each, 30 steps)
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region" | sed 's/.*timed region done: //'; }
(run A=1; run MPV_BAND_MAXR=8; run MPV_BAND_MAXR=12; run MPV_GEMM_BANDS=0; run MPV_BAND_REL192=78 MPV_BAND_REL160=70; run MPV_BAND_REL192=84 MPV_BAND_REL160=78; run MPV_BAND_THR=100; run A=1) 2>&1 | tee gpurun_out/r05_c13_band_knobs.log
