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
