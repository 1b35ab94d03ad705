#!/usr/bin/env bash
# Data-parallel training on the GPUs of one MI355X node: one process per GPU, RCCL over xGMI (the launch line the reference's
# scripts/train.sh:11-14 gives, on torch.distributed.run).  NUM_GPUS / MODEL_DIR / CONFIG can be set from the environment; further
# `key value` pairs are config overrides, e.g. `scripts/train.sh train.eval_interval_epoch 20`.
set -euo pipefail
cd "$(dirname "$0")/.."
NUM_GPUS=${NUM_GPUS:-$(python -c 'import torch; print(max(torch.cuda.device_count(), 1))')}
CONFIG=${CONFIG:-baseline.hrnetw32}
MODEL_DIR=${MODEL_DIR:-./log/rssformer}
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC: RCCL and the peer-to-peer SyncBN windows need it on this driver
python -m torch.distributed.run --nnodes=1 --nproc-per-node "${NUM_GPUS}" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-9696}" \
    train.py --config_path="${CONFIG}" --model_dir="${MODEL_DIR}" "$@"
