set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -m pytest tests/test_train_parity.py tests/test_autograd_bridge.py -q -m gpu -s -k "gradients_match or base_model or f8 or wgrad or attention_bwd or layernorm_bwd or two_rank" 2>&1 | grep -E "grad parity|passed|failed|worst" > $OUT/r03_n_grad_parity.txt
cat $OUT/r03_n_grad_parity.txt
rocprofv3 -L 2>/dev/null | grep -iE "TCC_EA0_RDREQ|DRAM|MALL|TCC_EA0_WRREQ|HBM" | head -40 > $OUT/r03_n_counters.txt
cat $OUT/r03_n_counters.txt | cut -c1-200
