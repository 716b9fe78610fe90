#!/bin/bash
# r05 call 1: first GPU contact of tools/probe/g256p_probe.hip + production GEMM beside it + a bench line of the untouched tree (box reference)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/probe/run_g256p.sh gpurun_out/r05_c1_g256p_first_contact.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_c1_bench.log 2>&1
tail -2 gpurun_out/r05_c1_bench.log
