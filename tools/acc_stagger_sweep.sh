# accurate-mode forward under different phase-stagger settings of the persistent GEMM (run on the GPU box)
python tools/accurate_fwd.py
for pct in 8 14 30; do echo "PCT $pct"; SF_G256_STAGGER_PCT=$pct python tools/accurate_fwd.py; done
for g in 2 4 6; do echo "GROUPS $g"; SF_G256_STAGGER_GROUPS=$g python tools/accurate_fwd.py; done
echo "PCT 10 GROUPS 4"; SF_G256_STAGGER_PCT=10 SF_G256_STAGGER_GROUPS=4 python tools/accurate_fwd.py
python tools/accurate_fwd.py
