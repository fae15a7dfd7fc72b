cd $GRAFT_REPO_ROOT
B="--steps 20 --warmup 4 --no-cpu-baseline --no-extra --gpu-baseline off"
S='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"] or {}; print(d["value"], d["ms_per_step"], "host", d.get("host_issue_ms_per_step"), "frac", r.get("frac"), "as_run", (r.get("as_run") or {}).get("frac"))'
for m in 0 1 3; do
  rm -f /tmp/pp_$m.db
  PNSFM_PP=$m PNSFM_TUNE_DB=/tmp/pp_$m.db python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off --no-prof > /dev/null 2>&1
done
for i in 1 2; do for m in 0 1 3; do
  echo "PNSFM_PP=$m: $(PNSFM_PP=$m PNSFM_TUNE_DB=/tmp/pp_$m.db python bench.py $B 2>/dev/null | tail -1 | python -c "$S")"
done; done
