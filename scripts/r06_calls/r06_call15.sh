#!/bin/bash
# r06 call 15: one box, alternating: the tree as shipped vs the tree with round 6's two step-level changes switched off (composed projection per
# block, q scaled by the attention kernels) -- what the round moved, free of the box lottery
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
run() { python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do echo -n "round 6 as shipped            : "; run; echo -n "round-5 behaviour (knobs off) : "; MPV_VIT_COMPOSE_GROUP=0 MPV_VIT_PRESCALE_Q=0 run; done) 2>&1 | tee $OUT/r06_c15_round6_vs_knobs_off.log
