#!/bin/bash
mkdir -p gpurun_out/r4e
python tools/ab_lib.py --rounds 3 fastvocoder_amd/libfastvocoder_hip.so@nomerge fastvocoder_amd/libfastvocoder_hip.so 2>&1 | tee gpurun_out/r4e/ab.log
python tools/ab_lib.py --rounds 2 --batch 16 fastvocoder_amd/libfastvocoder_hip.so@nomerge fastvocoder_amd/libfastvocoder_hip.so 2>&1 | tee -a gpurun_out/r4e/ab.log
