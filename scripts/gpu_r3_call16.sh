#!/bin/bash
# round 3, GPU call 16: A/B on one box -- BPRMF adagrad epoch with the fused (user, coefficient) gather list (the committed library)
# against the library right before it (scripts/micro/variants, not committed); then the parity files of the gather paths on the
# cleaned-up code (legacy branch removed).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c16; mkdir -p $O
export TMPDIR=/tmp
cp buffalo_amd/libbuffalo_hip.so /tmp/libA.so
for r in 1 2; do
  cp /tmp/libA.so buffalo_amd/libbuffalo_hip.so; timeout 200 python scripts/bench_extra.py bpr_adagrad 2>&1 | grep "^bpr_adagrad" | sed "s/^/fused    /" | cut -c1-200
  cp scripts/micro/variants/libbuffalo_hip_prefused.so buffalo_amd/libbuffalo_hip.so; timeout 200 python scripts/bench_extra.py bpr_adagrad 2>&1 | grep "^bpr_adagrad" | sed "s/^/prefused /" | cut -c1-200
done 2>&1 | tee $O/adagrad_ab.txt
cp /tmp/libA.so buffalo_amd/libbuffalo_hip.so
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_bpr_gpu.py tests/test_comm_gpu.py tests/test_comm_ranks_gpu.py tests/test_residency_gpu.py tests/test_trained_models_ref.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=|Error" $O/pytest.log | tail -5
