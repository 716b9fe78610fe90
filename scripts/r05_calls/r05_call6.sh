#!/bin/bash
# r05 call 6: (a) g256p with whole-line DMA requests (timing-only attribution build) beside the real kernel; (b) the round's new / changed
# GPU tests (B = 32 oracle parity, bf16-stream mode at full depth, whole-tensor gradient goldens, plain-number gates, C-ABI error paths,
# parked-derivative cross-kernel check); (c) the bench line with the step-level roofline fraction and the clock / power sampler
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c6_g256p_full_lines.log; : > $OUT
cd tools/probe
for v in "-DABL_FULLLINE" ""; do
  n=g256p_$(echo "x$v" | sed 's/-D//g; s/ /_/g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -o /tmp/$n g256p_probe.hip 2>&1 | grep -A3 "error" | head -8
  for shape in "50432 768 768" "50432 2304 768" "50432 768 3072" "5120 2048 8192"; do
    echo "== $n $shape" | tee -a $OUT
    timeout 60 /tmp/$n $shape 2>&1 | tail -2 | tee -a $OUT
  done
done
cd $GRAFT_REPO_ROOT
export MPV_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_c6_parity.txt; rm -f $MPV_PARITY_REPORT
timeout 1500 python -m pytest -q -x -s -m gpu tests/test_parity_fullsize_gpu.py "tests/test_model_gpu.py::test_tiny_vs_reference_golden" \
  "tests/test_model_gpu.py::test_bf16_residual_stream_mode_vs_reference_golden" "tests/test_model_gpu.py::test_configA_vs_reference_golden" \
  "tests/test_kernels_gpu.py::test_cabi_error_paths_on_the_device" "tests/test_gemm256_gpu.py::test_parked_gelu_derivative_does_not_depend_on_the_tile_kernel" \
  "tests/test_entrypoint_gpu.py" > gpurun_out/r05_c6_tests.log 2>&1
tail -15 gpurun_out/r05_c6_tests.log
grep "gate " gpurun_out/r05_c6_tests.log | head -40
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_c6_bench.json 2> gpurun_out/r05_c6_bench.err
python - <<'P'
import json
r = json.load(open("gpurun_out/r05_c6_bench.json"))
ro = r["roofline"]
print("bench: ms/step", r["ms_per_step"], "value", r["value"], "frac", ro["frac"], "gemm_frac", ro["gemm_frac"], "gemm ms", ro["gemm_ms_per_step"],
      "sclk", ro.get("sclk_mhz"), ro.get("sclk_mhz_min"), "power", ro.get("power_w"), ro.get("power_w_max"), "samples", ro.get("clock_power_samples"), ro.get("clock_power_source"))
P
