#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_parity_fullsize_gpu.py tests/test_entrypoint_gpu.py -m gpu -q -k "dropout or own_masks or train_mode or ln_stream or graph or entrypoint_three or bit_identical" -p no:cacheprovider 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|Warning" | tail -6
