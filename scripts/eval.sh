#!/usr/bin/env bash
# Evaluation of one checkpoint on one GPU (reference scripts/eval.sh): mIoU table + palette PNGs.  `TTA=1` adds the flip / scale ensemble.
set -euo pipefail
cd "$(dirname "$0")/.."
CKPT=${CKPT:-./log/hrnetw32.pth}
CONFIG=${CONFIG:-baseline.hrnetw32}
python eval.py --ckpt_path="${CKPT}" --config_path="${CONFIG}" ${TTA:+--tta True} "$@"
