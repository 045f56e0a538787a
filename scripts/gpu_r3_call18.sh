#!/bin/bash
# round 3, GPU call 18: heavy users spread over all eight queues (im_user_hybrid=1) against over as few as needed (=2); the new
# conflict-free cases of the replica path.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c18; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do for h in 2 1 0; do
  timeout 200 python bench.py --steps 60 --warmup 5 --no-extra --no-cpu-baseline --mode im_user_hybrid=$h 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('im_user_hybrid=$h', 'ms_per_step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'], 'incl_merge %.3f' % d['roofline']['frac_incl_merge_kernels'])"
done; done 2>&1 | tee $O/hybrid_ab.txt
timeout 600 python -m pytest tests/test_bpr_gpu.py -k "conflict_free" -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=|Error|assert" $O/pytest.log | tail -8
