#!/bin/bash
# r05 calls 25-28: the driver's command on four more boxes of the pool (final tree): ms per step beside the shader clock and power of the timed region
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print('box', r['ms_per_step'], 'ms', r['value'], 'samples/s frac', ro['frac'], 'gemm_frac', ro['gemm_frac'], 'sclk', ro.get('sclk_mhz'), 'min', ro.get('sclk_mhz_min'), 'W', ro.get('power_w'), 'max', ro.get('power_w_max'))" | tee -a gpurun_out/r05_c25_boxes.log
