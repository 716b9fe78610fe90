#!/bin/bash
# r06 call 18: the ViT dQ kernel with the next key tile's score / dP products issued ahead of the current tile's softmax arithmetic (a build-time arm,
# tools/probe/bin/libmpv_dqpipe.so): attention tests on that library, then whole-step alternation and the kernel's time in a trace
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
MPV_LIB_PATH=tools/probe/bin/libmpv_dqpipe.so timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "attention_fwd_bwd or producer" 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do echo -n "tree    : "; run; echo -n "dq pipe : "; MPV_LIB_PATH=tools/probe/bin/libmpv_dqpipe.so run; done) 2>&1 | tee $OUT/r06_c18_dq_pipe_ab.log
cd /tmp; export TMPDIR=/tmp MPV_WGRAD_STREAM=0
for arm in tree pipe; do
  rm -rf /tmp/kt_$arm
  if [ $arm = pipe ]; then export MPV_LIB_PATH=$GRAFT_REPO_ROOT/tools/probe/bin/libmpv_dqpipe.so; else unset MPV_LIB_PATH; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$arm -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt_$arm -name "*.db" | head -1) /tmp/kt_$arm.md > /dev/null
  echo "$arm: $(grep dq_duo96 /tmp/kt_$arm.md)" | tee -a $OUT/r06_c18_dq_pipe_ab.log
done
