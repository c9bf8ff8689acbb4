#!/bin/bash
# Round 6, first GPU pass: new tests, the MFMA shape-hazard experiment, the bench line, per-step kernel tables
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_cnn.py -x -q -m gpu > gpurun_out/r6_t1.log 2>&1; echo "cnn tests rc=$?"
tail -3 gpurun_out/r6_t1.log
DVT_LIB_PATH=$R/tools/_bin/libdvt_hip_nofence.so timeout -k 10 300 python -m pytest tests/test_gpu_cnn.py -q -m gpu -k "temporal_forward_from_lds" > gpurun_out/r6_nofence.log 2>&1; echo "nofence rc=$?"
tail -15 gpurun_out/r6_nofence.log
timeout -k 10 600 python bench.py --detail-out gpurun_out/r6_bench_detail1.json > gpurun_out/r6_b1.log 2>&1; echo "bench rc=$?"
tail -c 3500 gpurun_out/r6_b1.log
