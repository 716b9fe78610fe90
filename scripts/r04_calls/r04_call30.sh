#!/bin/bash
# the bf16 residual-stream mode of the decoder (MPV_DECODER_STREAM=bf16) against the reference goldens
python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "bf16_residual_stream or tiny_vs_reference" 2>&1 | grep -E "passed|failed|Error|assert" | head
