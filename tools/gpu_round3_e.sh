#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_vivit.py tests/test_gpu_frame_transformer.py tests/test_gpu_dp.py -q -m gpu -x > $O/r3_t10.log 2>&1; echo "tests rc=$?"; tail -3 $O/r3_t10.log
bash tools/gpu_prof_stats.sh
