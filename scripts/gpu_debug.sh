export TMPDIR=/tmp
mkdir -p gpurun_out
./scripts/micro/atomics > gpurun_out/micro_atomics.log 2>&1; cat gpurun_out/micro_atomics.log
