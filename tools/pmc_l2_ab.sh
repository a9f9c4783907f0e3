R=$PWD; cd /tmp; export TMPDIR=/tmp
for opt in "" "L2S_OPT=skinny_rc_jb=2"; do
  rm -rf /tmp/p_l2; env $opt timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/p_l2 -o c -- python $R/tools/prof_decode.py > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  echo "== $opt"; python $R/tools/pmc_read.py $(find /tmp/p_l2 -name "*.db" | head -1) "%skinny_rc%4, 2, %"
done
