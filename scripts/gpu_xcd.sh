# Policy-2 (per-XCD replicas) pass: knob sweep + quality study first, then the whole GPU suite.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/xcd
rm -rf $O; mkdir -p $O
rm -f $R/gpurun_out/xcd_study.json
timeout 700 python scripts/xcd_study.py timing planted ml20m > $O/study.log 2>&1; echo "study rc=$?" >> $O/study.log
cp $R/gpurun_out/xcd_study.json $O/ 2>/dev/null
timeout 900 python -m pytest ${PYTEST_TARGETS:-tests} -m gpu -q --timeout 600 -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^(timing|planted|ml20m|study)" $O/study.log | cut -c1-400
tail -25 $O/pytest.log
