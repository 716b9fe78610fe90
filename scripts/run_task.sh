#!/usr/bin/env bash
# One launcher for the five entry points, in place of the reference's scripts/run_{caption,cls,retrieval,retrieval_itm}_gpt3_1.3b.sh
# and its pre-training command line (README "Pre-training"): one process per GPU under torch.distributed.run, RCCL over xGMI.
#
#   scripts/run_task.sh <pretrain|retrieval|retrieval_itm|cls|caption> <config.yaml> <output_dir> [entry-point flags ...]
#
#   NPROC (default: all visible GPUs), MASTER_ADDR (127.0.0.1), MASTER_PORT (29500), WORLD_SIZE = nodes (1), RANK = node rank (0)
#   fine-tuning from a pre-trained checkpoint:   ... --resume path/to/1_3B_mp_rank_00_model_states.pt
#   evaluation only:                             NPROC=1 ... --evaluate_only --resume path/to/checkpoint
#   without the datasets (synthetic clips):      ... --synthetic_steps 20
set -euo pipefail
task=${1:?task}; config=${2:?config yaml}; out=${3:?output dir}; shift 3
root=$(cd "$(dirname "$0")/.." && pwd)
case "$task" in
  pretrain)      entry=$root/run_pretrain_distributed_gpt3.py ;;
  retrieval)     entry=$root/downstream/run_retrieval_distributed_gpt3.py ;;
  retrieval_itm) entry=$root/downstream/run_retrieval_distributed_gpt3_itm.py ;;
  cls)           entry=$root/downstream/run_cls_distributed_gpt3.py ;;
  caption)       entry=$root/downstream/run_caption_distributed_gpt3.py ;;
  *) echo "unknown task '$task'" >&2; exit 2 ;;
esac
nproc=${NPROC:-$(python -c 'import torch; print(max(torch.cuda.device_count(), 1))')}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0                 # dmabuf IPC: RCCL across processes needs it on this driver
PYTHONPATH=${PYTHONPATH:-}:$root python -m torch.distributed.run --nproc-per-node "$nproc" --nnodes "${WORLD_SIZE:-1}" --node-rank "${RANK:-0}" \
  --master-addr "${MASTER_ADDR:-127.0.0.1}" --master-port "${MASTER_PORT:-29500}" \
  "$entry" --config "$config" --output_dir "$out" --enable_deepspeed --bf16 "$@" 2>&1 | tee "$out/train.log"
