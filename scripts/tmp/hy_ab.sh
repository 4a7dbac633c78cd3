cd /tmp; export TMPDIR=/tmp
for e in 0 1 2 3; do
  rm -rf /tmp/hyp; PYGIM_TUNE=lds_code_exp=$e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hyp -o hy -- python3 $GRAFT_REPO_ROOT/scripts/exp_hybrid.py --kinds sbm --min 64 --reps 6 > /tmp/hy_$e.log 2>&1
  f=$(find /tmp/hyp -name "*kernel_stats.csv" | head -1)
  echo "exp=$e: $(grep k_lds_code8 $f | cut -d, -f2-4)"; grep "lds_hybrid=1" /tmp/hy_$e.log | cut -c1-120
done
