#!/bin/bash
# r05 call 2: attribution of the g256p K-step (ablation builds: no DMA / no fragment reads / no barrier / one m0 write per operand half)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_c2_g256p_ablation.log; : > $OUT
cd tools/probe
for v in "" "-DABL_M0X4" "-DABL_NODMA" "-DABL_NOREAD" "-DABL_NOBAR" "-DABL_NODMA -DABL_NOREAD" "-DABL_NODMA -DABL_NOREAD -DABL_NOBAR"; do
  n=g256p_abl$(echo "$v" | sed 's/-DABL//g; s/ //g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -o /tmp/$n g256p_probe.hip 2>/dev/null
  for shape in "50432 768 3072" "50432 2304 768" "50432 768 768"; do
    echo "== $n $shape" | tee -a ../../$OUT
    timeout 60 /tmp/$n $shape 2>&1 | tail -2 | tee -a ../../$OUT
  done
done
# discover the clock / power sources bench.py's sampler can use on this box
cd $GRAFT_REPO_ROOT
python - <<'P' 2>&1 | tee -a gpurun_out/r05_c2_sampler_probe.log
import glob, sys
sys.path.insert(0, '.')
import bench
s = bench.ClockPowerSampler(0)
print("source", s.source, "read", s._read() if s._read else None)
print(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/*")[:40])
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print({k: v for k, v in m.items() if 'clk' in k or 'power' in k or 'energy' in k})
    print(amdsmi.amdsmi_get_power_cap_info(h))
except Exception as e:
    print("amdsmi failed", repr(e))
P
