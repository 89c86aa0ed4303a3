#!/bin/bash
# Collects the per-round measurement artefacts on the GPU box (run via gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01b'
# then copy gpurun_out/profiles_<tag>/* into profiles/.  PMC passes run separately from the kernel trace (see the pool rule).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -q 2>&1 | tail -3 > "$OUT/${TAG}_pytest_gpu.log"
python bench.py --no-cpu-baseline --stage-profile "$OUT/${TAG}_stages.json" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats && mkdir -p /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc_$C && mkdir -p /tmp/prof_pmc_$C
  rocprofv3 --pmc $C -d /tmp/prof_pmc_$C -o p -- python "$ROOT/tools/prof_embed.py" 128 2 > /dev/null 2>&1
done
python - "$OUT/${TAG}_pmc_hbm.json" <<'EOF'
import collections, glob, json, sqlite3, statistics, sys
def per_kernel(counter):
    db = glob.glob("/tmp/prof_pmc_%s/**/*.db" % counter, recursive=True)[0]
    c = sqlite3.connect(db)
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for disp, name, cn, val in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        if cn == counter:
            d[name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][disp] += val
    return {k: list(v.values()) for k, v in d.items()}
f, w = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
out = {}
for k in f:
    if not k.startswith("conv"):
        continue
    fm, wm = statistics.median(f[k]), statistics.median(w.get(k, [0]))
    # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read)
    out[k] = {"launches": len(f[k]), "fetch_size_kb_median": fm, "write_size_kb_median": wm, "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump({"per_kernel": out,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace domains) over tools/prof_embed.py 128 2; median over "
                   "all launches of a kernel symbol; FETCH_SIZE doubled per MI355X_MICROARCH.md; WRITE_SIZE uncorrected (uncalibrated)"},
          open(sys.argv[1], "w"), indent=1)
EOF
cp "$OUT/${TAG}_pmc_hbm.json" "$ROOT/profiles/" 2>/dev/null   # bench.py reads the newest profiles/*_pmc_hbm.json for roofline.traffic
cd "$ROOT"
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.stderr"
ls -la "$OUT"
