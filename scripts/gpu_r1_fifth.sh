set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scripts/debug_racy.py > gpurun_out/debug_racy.log 2>&1
timeout 900 python -m pytest tests/test_bpr_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_bpr.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bpr.log
: > gpurun_out/sweep4_r1.jsonl
for m in "" "--mode hogwild_atomic=0" "--mode hogwild_atomic=0 --mode prefetch=0" "--mode hogwild_atomic=1" "--mode hot_items=64" "--mode hot_items=2048" "--mode hogwild_atomic=0 --mode waves_per_cu=16" "--mode hogwild_atomic=0 --mode chunk=1024"; do
  echo "## $m" >> gpurun_out/sweep4_r1.jsonl
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $m >> gpurun_out/sweep4_r1.jsonl 2>> gpurun_out/sweep4_r1.err
done
timeout 900 python scripts/quality_study.py > gpurun_out/quality_study.log 2>&1
cat gpurun_out/debug_racy.log | tail -9
tail -12 gpurun_out/pytest_bpr.log; python - <<'PY'
import json
for l in open("gpurun_out/sweep4_r1.jsonl"):
    if l.startswith("##"): print(l.strip()); continue
    try:
        j=json.loads(l); print("   ms/epoch %.2f  updates/s %.3g  frac %.3f"%(j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e: print("   ?", l[:100])
PY
tail -8 gpurun_out/quality_study.log; tail -3 gpurun_out/sweep4_r1.err
