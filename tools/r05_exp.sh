#!/bin/bash
# scratch: pairing of consecutive calls (frt_pipeline_set_pairing) - tests, 4-frame step with and without, the proxy in the default line
set -u
TAG=${1:-r05v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_coalesce.py tests/test_gpu_headline.py tests/test_cpp_shells.py -q -x 2>&1 | tail -15 > "$OUT/${TAG}_pytest.log"
for i in 1 2; do
python bench.py --batch 4 --no-cpu-baseline --steps 300 --no-extras > "$OUT/${TAG}_bench_b4_$i.json" 2>/dev/null
python bench.py --batch 4 --no-cpu-baseline --steps 300 --no-extras --pair > "$OUT/${TAG}_bench_b4_pair_$i.json" 2>/dev/null
python bench.py --batch 8 --no-cpu-baseline --steps 300 --no-extras > "$OUT/${TAG}_bench_b8_$i.json" 2>/dev/null
done
python bench.py --faces 1 --no-cpu-baseline --steps 100 --no-extras > "$OUT/${TAG}_bench_k1.json" 2>/dev/null
python bench.py --faces 1 --no-cpu-baseline --steps 100 --no-extras --pair > "$OUT/${TAG}_bench_k1_pair.json" 2>/dev/null
python bench.py --no-cpu-baseline --steps 50 > "$OUT/${TAG}_bench.json" 2>"$OUT/${TAG}_bench.stderr"
python - "$OUT" "$TAG" <<'PY'
import json,sys,glob,os
out,tag=sys.argv[1:3]
for f in sorted(glob.glob(os.path.join(out,tag+"_bench*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"unreadable"); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], json.dumps(d.get("strong_scaling_proxy",{}))[:600] if "strong_scaling_proxy" in d else "")
PY
cat "$OUT/${TAG}_pytest.log"
