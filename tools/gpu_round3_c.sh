#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
python -m pytest tests/test_gpu_vivit.py tests/test_gpu_pyramid.py -q -m gpu -s > $O/r3_t6.log 2>&1; echo "vivit+pyramid rc=$?"; tail -8 $O/r3_t6.log
python -m pytest tests/test_gpu_cnn.py -q -m gpu -s -k "default_frame or r2plus1d" > $O/r3_t7.log 2>&1; echo "cnn rc=$?"; tail -8 $O/r3_t7.log
