set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -m pytest tests/test_train_parity.py -q -m gpu --tb=short -k "rccl" 2>&1 | tail -30 > $OUT/r03_b_rccl.log
for B in 1 2 4; do python tools/fwd_time.py $B 16; done > $OUT/r03_b_fwd_time.txt 2>&1
SF_MODE=fp32 python tools/fwd_time.py 1 16 >> $OUT/r03_b_fwd_time.txt 2>&1
python tools/fwd_time.py 8 16 >> $OUT/r03_b_fwd_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for B in 1 2; do
rocprofv3 --kernel-trace -d /tmp/prof_b$B -o x -- python $R/tools/b1_trace.py $B > /dev/null 2>&1
python $R/profiles/summarize.py $(find /tmp/prof_b$B -name "*.db" | head -1) > $OUT/r03_base_b${B}_kernel_stats.txt
done
cat $OUT/r03_b_fwd_time.txt; tail -5 $OUT/r03_b_rccl.log
