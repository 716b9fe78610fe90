#!/bin/bash
# r06 call 12: the last tree after the closing set (bench.py's memory guard in front of the self-check): the distributed branch forced at world 1 on
# configs B and E, the driver's command, the entry-point and graph tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
for c in B E; do
  MPV_BENCH_FORCE_DIST=1 python bench.py --config $c --no-cpu-baseline --steps 10 > $OUT/r06_c12_forced_dist_$c.json 2> $OUT/r06_c12_forced_dist_$c.err
  grep -E "self-check|timed" $OUT/r06_c12_forced_dist_$c.err | cut -c1-400
  python -c "import json;r=json.load(open('$OUT/r06_c12_forced_dist_$c.json'));print('$c',r['ms_per_step'],r['step_mode'],r['host']['graph_self_check'],r['host']['mode_probe_ms_per_step'])"
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r06_c12_bench_B.json 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r06_c12_bench_B.json'));print(r['ms_per_step'],r['value'],r['step_mode'],r['roofline']['frac'],r['roofline']['traffic'],r['roofline']['sclk_mhz'])"
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_entrypoint_gpu.py tests/test_model_gpu.py -k "entrypoint or graph or bench" 2>&1 | tail -3
