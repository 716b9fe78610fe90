#!/bin/bash
# new test: bias slices of any alignment through the 256x256 kernel's LDS-DMA fetch / the 128x128 fallback
python -m pytest tests/test_gemm256_gpu.py -m gpu -q -p no:cacheprovider -k "bias_slice" 2>&1 | grep -E "passed|failed|Error|assert" | head
