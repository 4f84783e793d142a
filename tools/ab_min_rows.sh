R=$GRAFT_REPO_ROOT; cd $R
show='import sys,json; j=json.loads(sys.stdin.read()); q=j.get("qres34m") or {}; print(sys.argv[1], j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"], "qres34m", q.get("value"))'
for i in 1 2 3; do
  for rows in 49152 24576 6144; do
    python tools/bench_min_rows.py $rows --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 2>/dev/null | python -c "$show" min_rows=$rows
  done
done
