export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python scripts/bench_extra.py eals ) 2>&1 | grep -E "^eals|real|Error" | cut -c1-330
