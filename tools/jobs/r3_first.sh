#!/bin/bash
# round 3, first GPU call: the GPU test suite and the default bench line of the new bench.py
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
nproc > gpurun_out/r3a/host.txt; free -g >> gpurun_out/r3a/host.txt; df -h /dev/shm >> gpurun_out/r3a/host.txt; rocm-smi --showmeminfo vram >> gpurun_out/r3a/host.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r3a/pytest.log 2>&1; tail -5 gpurun_out/r3a/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err; tail -c 600 gpurun_out/r3a/bench_default.err
head -c 1500 gpurun_out/r3a/bench_default.json
