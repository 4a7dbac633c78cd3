#!/bin/bash
# round-4 evidence run on the GPU box: config / inference tables, MFMA block densities; results under gpurun_out/r04/
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04
timeout 900 python scripts/exp_configs.py --cases reddit:CSR:f32:256,reddit:COO:i32:256,reddit:CSR:i32:256,reddit:CSR:i16:256,reddit:CSR:i8:256,reddit:CSR:f64:256,reddit:CSR:i64:256,reddit:CSR:f32:128,reddit:CSR:f32:64,reddit:CSR:f32:100,reddit:CSR:i8:100,reddit:CSR:f64:100,ogbn-products:COO:i32:256,ogbn-products:CSR:f32:256 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/config_table.txt
bash scripts/inference_table.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/inference_table.txt
timeout 600 python scripts/exp_mfma_density.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/mfma_density.txt
