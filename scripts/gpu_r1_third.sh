set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_r1.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r1.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_r1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r1.log
: > gpurun_out/sweep2_r1.jsonl
for m in "" "--mode hot_items=128" "--mode hot_items=256" "--mode hot_items=1024" "--mode hot_items=2048" "--mode hogwild_atomic=0" "--mode hogwild_atomic=1"; do
  echo "## $m" >> gpurun_out/sweep2_r1.jsonl
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $m >> gpurun_out/sweep2_r1.jsonl 2>> gpurun_out/sweep2_r1.err
done
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
tail -3 gpurun_out/smoke_r1.log; tail -15 gpurun_out/pytest_r1.log; python - <<'PY'
import json
for l in open("gpurun_out/sweep2_r1.jsonl"):
    if l.startswith("##"): print(l.strip()); continue
    try:
        j=json.loads(l); print("   ms/epoch %.2f  updates/s %.3g  frac %.3f"%(j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e: print("   ?", l[:100])
PY
cat gpurun_out/bench_r1.json
