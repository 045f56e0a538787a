export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python scripts/bench_extra.py warp_c5 ) 2>&1 | grep -E "^warp_c5|real|Error|error" | cut -c1-330
