#!/bin/bash
# Round 6, after the LAST source edit (the counter file is stamped with a hash over csrc/ + include/): the ingest tests (host re-parse on threads), the
# rocprofv3 passes of the bench command, then the default bench line on the same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final2; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ingest_gpu.py tests/test_sppmi.py -q -m gpu > $O/ingest_tests.txt 2>&1; echo "ingest rc=$?"; tail -2 $O/ingest_tests.txt
PROF_DIR=r6prof2 bash scripts/gpu_profile.sh > $O/profile_stdout.txt 2>&1; echo "profile rc=$?"; tail -12 $O/profile_stdout.txt | cut -c1-200
cp gpurun_out/r6prof2/pmc_latest.json $O/ 2>/dev/null
cp gpurun_out/r6prof2/pmc_latest.json profiles/pmc_latest.json 2>/dev/null
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.out | wc -c; tail -1 $O/bench_default.out | cut -c1-1500
cp bench_extra.json $O/bench_extra_default.json 2>/dev/null
cp gpurun_out/r6prof2/bench_under_rocprof.json $O/ 2>/dev/null
for f in $(find gpurun_out/r6prof2/stats -name "*kernel_stats.csv"); do cp $f $O/bench_rocprofv3_kernel_stats.csv; done
