#!/bin/bash
# r05 call 11: measurement only -- the step's plain GEMM shapes through the vendor library (torch.matmul / F.linear) beside mpv_gemm_bf16, same box;
# + the data-parallel graph segments test after the empty-segment fix
cd $GRAFT_REPO_ROOT
timeout 600 python tools/gemm_vs_vendor.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_c11_gemm_vs_vendor.log
timeout 600 python -m pytest -q -x -m gpu -p no:cacheprovider "tests/test_model_gpu.py::test_graph_step_data_parallel_segments_forced_world1" "tests/test_model_gpu.py::test_graph_step_bit_identical_to_eager" 2>&1 | tail -4
