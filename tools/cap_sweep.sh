for cap in 2048 4096 8192; do for b in 1 2; do
  SF_SKINNY_MAX_M=$cap python bench.py --batch $b --no-cpu-baseline --no-train 2>/tmp/err.txt | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
if not t: print('cap $cap B $b: no output'); sys.exit()
d=json.loads(t[-1]); print('cap $cap B $b:', d['value'], d['ms_per_step'])"
  tail -2 /tmp/err.txt | grep -i "error\|Traceback\|rror" | head -2
done; done
