#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
python tests/probes/r2plus1d_grad_probe.py 4 32 > $O/r3_p1.log 2>&1; echo "r2plus1d probe rc=$?"
python tests/probes/r2plus1d_grad_probe.py 8 64 > $O/r3_p1b.log 2>&1; echo "r2plus1d probe (8x64) rc=$?"
python tests/probes/grad_stream_probe.py c2 > $O/r3_p2.log 2>&1; echo "grad stream probe rc=$?"
python tests/probes/grad_stream_probe.py metric > $O/r3_p3.log 2>&1; echo "grad stream probe metric rc=$?"
python -m pytest tests/test_gpu_vivit.py tests/test_gpu_pyramid.py -q -m gpu -s > $O/r3_t5.log 2>&1; echo "vivit+pyramid rc=$?"; tail -12 $O/r3_t5.log
