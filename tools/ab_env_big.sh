# same-box A/B of one environment switch at 384x1280 batch 2 (GPU-bound: host issue is ~45 % of the step there)
cd $GRAFT_REPO_ROOT
S='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"] or {}; print(d["value"], d["ms_per_step"], "host", d.get("host_issue_ms_per_step"), "frac", r.get("frac"), "as_run", (r.get("as_run") or {}).get("frac"))'
for i in 1 2; do for v in 0 1; do
  echo "$AB_VAR=$v: $(env $AB_VAR=$v python bench.py --height 384 --width 1280 --batch 2 --steps 12 --warmup 3 --no-cpu-baseline --no-extra --gpu-baseline off 2>/dev/null | tail -1 | python -c "$S")"
done; done
