#!/bin/bash
# Round 6, second GPU pass: full GPU suite, the MFMA shape-hazard experiment, per-step kernel tables (ALL rows) of the CNN workloads
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r6_t2.log 2>&1; echo "gpu tests rc=$?"
tail -8 gpurun_out/r6_t2.log
DVT_LIB_PATH=$R/tools/_bin/libdvt_hip_nofence.so timeout -k 10 300 python -m pytest tests/test_gpu_cnn.py -q -m gpu -k "temporal_forward_from_lds" > gpurun_out/r6_nofence.log 2>&1; echo "nofence rc=$?"
grep -c PASSED gpurun_out/r6_nofence.log; tail -8 gpurun_out/r6_nofence.log
for wl in frametransformer pyramid; do
cd /tmp && export TMPDIR=/tmp
out=r6_prof_$wl
rm -rf $R/gpurun_out/$out
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$out -- python3 $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' > $R/gpurun_out/$out.log 2>&1
f=$(find $R/gpurun_out/$out -name "*kernel_trace.csv" | head -1)
python3 $R/tools/dev/trace_steps.py $f 3 600 $R/gpurun_out/${out}_order.txt > $R/gpurun_out/$out.md 2>&1
rm -rf $R/gpurun_out/$out
head -4 $R/gpurun_out/$out.md
done
