#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_dp.py tests/test_gpu_vivit.py -q -m gpu -x > $O/r3_t8.log 2>&1; echo "ops+dp+vivit rc=$?"; tail -4 $O/r3_t8.log
python -m pytest tests/test_gpu_cnn.py tests/test_gpu_pyramid.py tests/test_gpu_frame_transformer.py -q -m gpu -x > $O/r3_t9.log 2>&1; echo "cnn+pyramid+ft rc=$?"; tail -4 $O/r3_t9.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r3_b3.log 2>&1; echo "bench rc=$?"; tail -c 600 $O/r3_b3.log
