set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
(rocminfo | grep -E "Name|Compute Unit|Max Clock" | head -20; rocm-smi --showmeminfo vram | head; nproc; lscpu | grep "Model name") > gpurun_out/box_r1.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_r1.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r1.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_r1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r1.log
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; echo "bench rc=$?" >> gpurun_out/bench_r1.err
tail -5 gpurun_out/smoke_r1.log; tail -40 gpurun_out/pytest_r1.log; cat gpurun_out/bench_r1.json; tail -5 gpurun_out/bench_r1.err
