cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/test_gpu_all.log 2>&1; echo "all rc=$?" >> gpurun_out/test_gpu_all.log
tail -4 gpurun_out/test_gpu_all.log
timeout 600 python spmm_test.py --dataset Reddit --version spmm --sp_format CSR --data_type FLT32 --lib_path ./backend_pim/spmm_default/build/libbackend_pim.so 2>&1 | grep -a "DATA\|rror" | tail -6
timeout 600 python spmm_test.py --dataset Reddit --version spmm --data_type INT32 --lib_path ./backend_pim/spmm_default/build/libbackend_pim.so 2>&1 | grep -a "DATA\|rror" | tail -4
timeout 300 python inference.py --dataset Reddit --num_layers 3 --hidden_size 256 --version spmm --lib_path ./backend_pim/spmm_default/build/libbackend_pim.so --model gcn --data_type FLT32 --repeat 5 --graph 1 2>/dev/null | grep infer_time | sort -t: -k2 -n | head -1
timeout 300 python inference.py --dataset Reddit --num_layers 3 --hidden_size 256 --version spmm --lib_path ./backend_pim/spmm_default/build/libbackend_pim.so --model gcn --data_type INT16 --repeat 5 2>/dev/null | grep infer_time | sort -t: -k2 -n | head -1
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log | cut -c1-300
