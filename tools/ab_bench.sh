# A/B of run-time options through the timed region of bench.py itself, on the diagnostic build (L2S_LIB=diag: the block-form switches of include/l2s_diag.h exist there only): bash tools/ab_bench.sh "<bench flags>" "<bench flags>" ...  -> profiles/rNN_ab_*.txt
for cfg in "$@"; do
  echo "== $cfg"
  L2S_LIB=diag python bench.py --skip-cpu-baseline --skip-train-leg $cfg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
done
