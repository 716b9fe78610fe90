#!/bin/bash
# final tree: whole GPU suite + smoke + the driver's bench command, configs D and E side lines
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r04_final2_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/r04_final2_gpu_tests.log
grep -E "passed|failed|pytest rc" $OUT/r04_final2_gpu_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r04_final2_smoke.log 2>&1; tail -1 $OUT/r04_final2_smoke.log
timeout 600 python bench.py > $OUT/r04_final2_bench_B_1gpu.json 2> $OUT/r04_final2_bench_B_1gpu.err
timeout 300 python bench.py --config D --no-cpu-baseline > $OUT/r04_final2_bench_D.json 2> $OUT/r04_final2_bench_D.err
timeout 400 python bench.py --config E --no-cpu-baseline --steps 10 > $OUT/r04_final2_bench_E.json 2> $OUT/r04_final2_bench_E.err
python - <<PY
import json
for c in ["B_1gpu","D","E"]:
    try:
        r=json.load(open("$OUT/r04_final2_bench_%s.json"%c))
        print(c, "ms/step", r["ms_per_step"], "value", r["value"], "frac", r["roofline"]["frac"], "step_frac", r["roofline"].get("step_frac"), "traffic", r["roofline"].get("traffic"))
    except Exception as e: print(c, "ERR", e)
PY
