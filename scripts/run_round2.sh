cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gnn_gpu.py tests/test_wrappers_gpu.py tests/test_lds_gpu.py tests/test_fullsize_gpu.py -q -m gpu > gpurun_out/test_deq.log 2>&1; echo "rc=$?" >> gpurun_out/test_deq.log
tail -8 gpurun_out/test_deq.log
T=gpurun_out/inference_table.txt
: > $T
for m in gcn sage gin; do for dt in FLT32 INT32 INT8; do
  echo -n "$m $dt " >> $T
  timeout 300 python inference.py --dataset Reddit --num_layers 3 --hidden_size 256 --version spmm --lib_path ./backend_pim/spmm_default/build/libbackend_pim.so --model $m --data_type $dt --repeat 5 2>/dev/null | grep infer_time | sort -t: -k2 -n | head -1 >> $T
done; done
cat $T
timeout 1200 python scripts/exp_configs.py --cases "reddit:CSR:f32:256,reddit:COO:i32:256,reddit:CSR:f64:256,reddit:CSR:i32:100,reddit:CSR:i8:256,reddit:CSR:i16:256,reddit:CSR:f32:128,reddit:CSR:f32:64,reddit:CSR:f32:32,reddit:CSR:f32:41,ogbn-products:COO:i32:256,ogbn-products:CSR:f32:256,ogbn-papers100M:CSR:f32:16,ogbn-papers100M:CSR:f32:32" > gpurun_out/config_table.txt 2>&1
grep -v amdgpu gpurun_out/config_table.txt
timeout 900 python scripts/exp_products.py > gpurun_out/exp_products.txt 2>&1
grep -v amdgpu gpurun_out/exp_products.txt
