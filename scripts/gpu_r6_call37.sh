#!/bin/bash
# GPU call 37 (round 6): the distribution of the bench line's value over processes on one box (the per-handle draw as the driver's single run meets it)
O=gpurun_out/r6c37; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('run $i: %.3f G updates/s, %.3f ms per step, kernel %.3f ms, frac %.3f, triad %.0f GB/s' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], r['frac'], r['triad_GBps']))"; done | tee $O/value_distribution.txt
