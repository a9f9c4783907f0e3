# A/B of run-time options through the timed region of bench.py itself: bash tools/ab_bench.sh "<bench flags>" "<bench flags>" ...  -> profiles/rNN_ab_*.txt
for cfg in "$@"; do
  echo "== $cfg"
  python bench.py --skip-cpu-baseline --skip-train-leg $cfg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
done
