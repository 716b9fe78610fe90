#!/bin/bash
# Scaling curve of the pre-train step on ONE node: bench.py at N = 1, 2, 4, 8 GPUs back to back (the driver's SCALE run does the same
# with its own clock), plus a kernel trace of rank 0 at every N >= 2 analysed for all-reduce / backward overlap (tools/rocpd_overlap.py)
# and a per-N power / clock sample.  No such run has been possible on the 1-GPU boxes this repository was built on: this script is the
# first thing to run when a node is available.   Usage: bash scripts/bench_scale.sh [steps] [warmup] [outdir]
set -u
STEPS=${1:-30}; WARM=${2:-5}; OUT=${3:-gpurun_out/scale}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -le "$NG" ] || { echo "only $NG GPUs: skipping N=$N"; continue; }
  PORT=$((29600 + N))
  if [ "$N" -eq 1 ]; then LAUNCH="python"; else LAUNCH="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT"; fi
  $LAUNCH "$R/bench.py" --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  tail -1 "$OUT/bench_n$N.json" | cut -c1-300
  if [ "$N" -ge 2 ]; then
    # the launcher runs under the profiler: one database per rank, the first one found is analysed
    rm -rf /tmp/kt_n$N
    ( cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_n$N -o n$N -- python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
        --master-addr 127.0.0.1 --master-port $((PORT + 50)) "$R/bench.py" --gpus $N --steps 3 --warmup 1 --no-cpu-baseline --no-roofline ) > "$OUT/trace_n$N.log" 2>&1
    DB=$(find /tmp/kt_n$N -name "*.db" | head -1)
    [ -n "$DB" ] && python "$R/tools/rocpd_overlap.py" "$DB" "$OUT/overlap_n$N.md" | tail -3
  fi
done
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
base = None
for n in (1, 2, 4, 8):
    p = os.path.join(out, f"bench_n{n}.json")
    if not os.path.isfile(p):
        continue
    try:
        rec = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(f"N={n}: no JSON line ({e})"); continue
    base = base or rec["value"]
    print(f"N={n}: {rec['value']:.1f} samples/s, {rec['ms_per_step']:.2f} ms/step, x{rec['value'] / base:.2f} of N=1")
PY
