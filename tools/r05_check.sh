#!/bin/bash
# final-tree check on the GPU box: smoke(), the whole GPU suite, the driver's bench command twice, detector traces at 32 / 4 / 1 frames
set -u
TAG=${1:-r06b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_smoke.txt"
python -m pytest tests -m gpu -q 2>&1 | tail -4 > "$OUT/${TAG}_pytest_gpu.log"
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_$i.json" 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
for B in 32 4 1; do
  rm -rf /tmp/prof_det && mkdir -p /tmp/prof_det
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o st -- python "$ROOT/tools/prof_det.py" $B 5 > /dev/null 2>&1
  cp "$(find /tmp/prof_det -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_det_kernel_stats_b$B.csv" 2>/dev/null
done
cd "$ROOT"
for f in "$OUT"/${TAG}_det_kernel_stats_b*.csv; do echo "== $(basename $f)"; python tools/det_table.py "$f"; done > "$OUT/${TAG}_det_tables.txt" 2>&1
cat "$OUT/${TAG}_smoke.txt" "$OUT/${TAG}_pytest_gpu.log"; grep "kernels per" "$OUT/${TAG}_det_tables.txt"
python - "$OUT" "$TAG" <<'PY'
import json,sys,glob,os
out,tag=sys.argv[1:3]
for f in sorted(glob.glob(os.path.join(out,tag+"_bench*.json"))):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; p=d.get("strong_scaling_proxy") or {}
    print(os.path.basename(f), d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], {k:(v if not isinstance(v,dict) else v.get("projected_x_at_8")) for k,v in p.items() if k in ("projected_x_at_8","paired","grouped_by_4")})
PY
