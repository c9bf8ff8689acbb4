#!/bin/bash
# one gpurun call: parity tests touched this round, the 2-rank launcher on one GPU, a kernel-stats profile of the bench
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
python -m pytest tests/test_gpu_vivit.py tests/test_gpu_pyramid.py -x -q -m gpu -s > $O/r3_t3.log 2>&1; echo "vivit+pyramid rc=$?"; tail -3 $O/r3_t3.log
python -m pytest tests/test_gpu_cnn.py -x -q -m gpu -s -k "default_frame or r2plus1d or frame_transformer" > $O/r3_t4.log 2>&1; echo "cnn rc=$?"; tail -3 $O/r3_t4.log
python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $O/r3_b2.log 2>&1; echo "2-rank gloo rc=$?"; tail -c 1500 $O/r3_b2.log
bash tools/run_profiles.sh > $O/r3_prof.log 2>&1; echo "profiles rc=$?"
python3 tools/profile_summarize.py $O/prof $O/prof_summary r03 > $O/r3_prof_sum.log 2>&1; echo "summary rc=$?"
