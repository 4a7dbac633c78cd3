cd $GRAFT_REPO_ROOT
timeout 300 python scripts/exp_lds.py quick 2>&1 | grep -v "^small n" | grep -v amdgpu | tail -3
bash scripts/lds_pmc.sh w16c --waves 16 --clustered > gpurun_out/lds_pmc_w16c.log 2>&1
grep -A40 "k_lds_spmm_f32_w16b" gpurun_out/ldspmc_w16c/summary.txt | head -42
bash scripts/profile_inference.sh --dataset Reddit --num_layers 3 --hidden_size 256 --version spmm --lib_path ./backend_pim/spmm_default/build/libbackend_pim.so --model gcn --data_type FLT32 --repeat 3 > gpurun_out/prof_inf_flt32.txt 2>&1
head -16 gpurun_out/prof_inf_flt32.txt | cut -c1-200
