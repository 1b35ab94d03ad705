#!/usr/bin/env bash
# Class-id PNGs for the test tiles of the config (reference scripts/predict_test.sh), one GPU.
set -euo pipefail
cd "$(dirname "$0")/.."
CKPT=${CKPT:-./log/hrnetw32.pth}
CONFIG=${CONFIG:-baseline.hrnetw32}
python predict.py --ckpt_path="${CKPT}" --config_path="${CONFIG}" --out_dir="${OUT_DIR:-./out}" "$@"
