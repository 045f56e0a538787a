# Hogwild-policy pass: BPR GPU tests, then the knob sweep + quality study (scripts/xcd_study.py).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/xcd
rm -rf $O; mkdir -p $O
rm -f $R/gpurun_out/xcd_study.json
timeout 600 python -m pytest ${PYTEST_TARGETS:-tests} -m gpu -q --timeout 180 -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 500 python scripts/xcd_study.py ${STUDY:-timing planted ml20m} > $O/study.log 2>&1; echo "study rc=$?" >> $O/study.log
cp $R/gpurun_out/xcd_study.json $O/ 2>/dev/null
grep -E "^(timing|planted|ml20m|study)" $O/study.log | cut -c1-420
tail -5 $O/study.log | cut -c1-300
if [ -n "$PMC_MODES" ]; then
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o p -- python $R/scripts/run_policy.py $PMC_MODES > $O/prof_stats.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/scripts/run_policy.py $PMC_MODES epochs=2 > $O/pmc_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/scripts/run_policy.py $PMC_MODES epochs=2 > $O/pmc_write.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python $R/scripts/run_policy.py $PMC_MODES epochs=2 > $O/pmc_sq.log 2>&1
  cd $R
  find $O -name "*kernel_trace.csv" -size +4M -delete
  grep -h run_policy $O/prof_stats.log $O/pmc_*.log | cut -c1-300
  for f in $(find $O/prof_stats -name "*kernel_stats.csv"); do head -12 $f; done
  python - <<PY
import csv, glob, collections
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            for cn, v in acc[k].items():
                if "bpr_" in k or "xcd_" in k:
                    print(d, k, cn, "sum %.4g" % v, "dispatches", n[(k, cn)], "per dispatch %.4g" % (v / n[(k, cn)]))
PY
fi
