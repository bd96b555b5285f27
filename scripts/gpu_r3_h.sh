#!/bin/bash
# round 3, call H: split IALS rows — the IALS parity suite, the sharded epoch test, the bench's IALS section
mkdir -p gpurun_out/h
timeout 150 python -m pytest tests/test_ials_gpu.py -x -q -m gpu > gpurun_out/h/pytest_ials.txt 2>&1; echo "ials rc=$?" >> gpurun_out/h/pytest_ials.txt
tail -5 gpurun_out/h/pytest_ials.txt
timeout 80 python -m pytest tests/test_sharding_gpu.py -x -q -m gpu -k "ials or IALS or two" > gpurun_out/h/pytest_shard.txt 2>&1; echo "shard rc=$?" >> gpurun_out/h/pytest_shard.txt
tail -3 gpurun_out/h/pytest_shard.txt
timeout 100 python scripts/ials_only.py > gpurun_out/h/ials.json 2> gpurun_out/h/ials.err; echo "ials section rc=$?"
tail -c 3000 gpurun_out/h/ials.json; tail -3 gpurun_out/h/ials.err
