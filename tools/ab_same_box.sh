#!/bin/bash
# Same-box A/B of the whole step, the only comparison that resolves < 2 % on this pool (boxes differ by +-4 %, repeated runs on one
# box agree to 0.1 ms).  Run inside ONE gpurun call.
#   bash tools/ab_same_box.sh env  MPV_VIT_COMPOSE 0 1          # an environment knob, alternating values, two rounds
#   bash tools/ab_same_box.sh lib  youku-mplug_amd/csrc/build/libmpv_old.so   # a previous build of the library against the tree's
# (build the old library first:  git stash; make -C youku-mplug_amd/csrc; cp youku-mplug_amd/libmpv_hip.so <path>; git stash pop; make ...;
#  <path> must be inside the tree and not under gpurun_out/ so that it travels to the GPU box)
set -u
MODE=${1:?env|lib}
run() { "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region" | sed 's/.*timed region done: //'; }
if [ "$MODE" = env ]; then
  VAR=$2; shift 2
  for round in 1 2; do for v in "$@"; do echo -n "$VAR=$v  "; run env $VAR=$v; done; done
else
  OLD=$2
  for round in 1 2; do echo -n "tree      "; run env; echo -n "$OLD  "; run env MPV_LIB_PATH=$OLD; done
fi
