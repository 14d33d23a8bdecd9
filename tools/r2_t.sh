#!/bin/bash
mkdir -p gpurun_out/tt
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "sweep_phases" > gpurun_out/tt/a.log 2>&1; tail -15 gpurun_out/tt/a.log | cut -c1-400
