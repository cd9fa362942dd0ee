#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 400 python tools/grad_error_breakdown.py 2>&1 | grep -v amdgpu.ids | tee $O/grad_breakdown.txt
