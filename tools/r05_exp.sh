#!/bin/bash
# scratch: groups of 2 / 4 consecutive calls per recogniser pass - tests, 4-frame step, the proxy in the default line
set -u
TAG=${1:-r05y}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_coalesce.py tests/test_cpp_shells.py -q -x 2>&1 | tail -15 > "$OUT/${TAG}_pytest.log"
for g in 0 2 3 4; do
python bench.py --batch 4 --no-cpu-baseline --steps 300 --no-extras --pair $g > "$OUT/${TAG}_bench_b4_pair$g.json" 2>/dev/null
done
python bench.py --faces 1 --no-cpu-baseline --steps 100 --no-extras --pair 4 > "$OUT/${TAG}_bench_k1_pair4.json" 2>/dev/null
python bench.py --batch 16 --no-cpu-baseline --steps 100 --no-extras > "$OUT/${TAG}_bench_b16.json" 2>/dev/null
python bench.py --no-cpu-baseline --steps 50 > "$OUT/${TAG}_bench.json" 2>"$OUT/${TAG}_bench.stderr"
python - "$OUT" "$TAG" <<'PY'
import json,sys,glob,os
out,tag=sys.argv[1:3]
for f in sorted(glob.glob(os.path.join(out,tag+"_bench*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"unreadable"); continue
    p=d.get("strong_scaling_proxy") or {}
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if a!="what"}) for k,v in p.items() if k!="note"})
PY
cat "$OUT/${TAG}_pytest.log"
