#!/bin/bash
# round 6, call E: device sharing, the reduction stand-in as the victim (tools/platform/add_victim.py), the RCCL world-size-1 collectives with their
# result line kept, and the synthetic victim beside the temporal backward with SF_TBWD_OWN_CU=1; then the bench line at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
{
echo "### reduction stand-in (torch.add fp32 + bf16 on a second stream) beside the real temporal attention backward, one process"
SF_VICTIM_SECONDS=25 timeout 200 python tools/platform/add_victim.py 2>&1 | grep "add victim"
echo "### the same with SF_TBWD_OWN_CU=1"
SF_TBWD_OWN_CU=1 SF_VICTIM_SECONDS=15 timeout 200 python tools/platform/add_victim.py 2>&1 | grep "add victim"
echo "### RCCL world-size-1 collectives beside the real temporal attention backward (another process)"
SF_NOISE_SECONDS=40 SF_NOISE_MODES=attn_bwd_temporal timeout 300 python tools/noise_ops.py > $OUT/e_noise1.log 2>&1 & NP=$!
for i in $(seq 1 200); do grep -q starting $OUT/e_noise1.log 2>/dev/null && break; sleep 1; done
SF_VICTIM_SECONDS=15 timeout 120 python tools/platform/rccl_victim.py 2>&1 | grep "rccl world"; wait $NP
echo "### synthetic victim beside the real temporal attention backward with SF_TBWD_OWN_CU=1 (another process)"
SF_TBWD_OWN_CU=1 SF_NOISE_SECONDS=40 SF_NOISE_MODES=attn_bwd_temporal timeout 300 python tools/noise_ops.py > $OUT/e_noise2.log 2>&1 & NP=$!
for i in $(seq 1 200); do grep -q starting $OUT/e_noise2.log 2>/dev/null && break; sleep 1; done
timeout 60 tools/platform/neighbor_lab victim 15 | grep -v "^  launch"; wait $NP
echo "### synthetic victim with a CU's whole LDS (163840 bytes) beside synthetic neighbour mode 3 (pure MFMA, no LDS): does owning the LDS help against a neighbour that needs none?"
timeout 60 tools/platform/neighbor_lab neighbour 3 16 & NP=$!
sleep 1; timeout 60 tools/platform/neighbor_lab victim 10 163840 | grep -v "^  launch"; wait $NP
} 2>&1 | tee $OUT/e_device_sharing.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "hidden_states or tower or nonfinite or lora_with_unfrozen or small_model_gradients" > $OUT/e_tests.log 2>&1; tail -4 $OUT/e_tests.log
timeout 600 python bench.py > $OUT/e_bench.json 2> $OUT/e_bench.err; tail -c 1500 $OUT/e_bench.json
