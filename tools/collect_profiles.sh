#!/bin/bash
# Collects the per-round measurement artefacts on the GPU box (run via gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02'
# then copy gpurun_out/profiles_<tag>/* into profiles/<round>/ (profiles/r06/ for tag r06z).  PMC passes run separately from the kernel trace (see the pool rule).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -q 2>&1 | tail -3 > "$OUT/${TAG}_pytest_gpu.log"
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace of the benchmark command itself
rm -rf /tmp/prof_stats && mkdir -p /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
# 2. counters, one group per pass, over a recogniser-only run: HBM bytes and matrix-pipe busy cycles
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  D=/tmp/prof_pmc_$(echo $C | cut -d' ' -f1)
  rm -rf $D && mkdir -p $D
  rocprofv3 --pmc $C -d $D -o p -- python "$ROOT/tools/prof_embed.py" 128 2 > /dev/null 2>&1
done
python - "$OUT/${TAG}_pmc_hbm.json" "$OUT/${TAG}_pmc_mfma.json" <<'PYEOF'
import collections, glob, json, sqlite3, statistics, sys
def per_kernel(dirname, counters):
    db = glob.glob("/tmp/prof_pmc_%s/**/*.db" % dirname, recursive=True)[0]
    c = sqlite3.connect(db)
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for disp, name, cn, val in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        if cn in counters:
            d[name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][cn][disp] += val
    return d
f, w = per_kernel("FETCH_SIZE", ["FETCH_SIZE"]), per_kernel("WRITE_SIZE", ["WRITE_SIZE"])
out = {}
for k in f:
    if not k.startswith("conv"):
        continue
    fm = statistics.median(f[k]["FETCH_SIZE"].values())
    wm = statistics.median(w.get(k, {}).get("WRITE_SIZE", {0: 0}).values())
    # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read)
    out[k] = {"launches": len(f[k]["FETCH_SIZE"]), "fetch_size_kb_median": fm, "write_size_kb_median": wm, "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump({"per_kernel": out,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace domains) over tools/prof_embed.py 128 2; median over "
                   "all launches of a kernel symbol; FETCH_SIZE doubled per MI355X_MICROARCH.md; WRITE_SIZE uncorrected (uncalibrated)"},
          open(sys.argv[1], "w"), indent=1)
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
m = per_kernel("SQ_VALU_MFMA_BUSY_CYCLES", names)
mo = {}
for k, v in m.items():
    if k.startswith("conv"):
        mo[k] = {cn: statistics.median(v[cn].values()) for cn in names if cn in v}
        mo[k]["launches"] = len(next(iter(v.values())))
json.dump({"per_kernel": mo, "note": "rocprofv3 --pmc (one pass) over tools/prof_embed.py 128 2, medians per launch, summed over the chip by rocprofv3; "
           "SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with the matrix pipe busy (1024 SIMDs), GRBM_GUI_ACTIVE is per-XCD busy cycles (8 XCDs)"},
          open(sys.argv[2], "w"), indent=1)
PYEOF
mkdir -p "$ROOT/profiles/${TAG:0:3}" && cp "$OUT/${TAG}_pmc_hbm.json" "$ROOT/profiles/${TAG:0:3}/" 2>/dev/null   # bench.py reads the newest profiles/*/*_pmc_hbm.json for roofline.traffic
cd "$ROOT"
# 3. the benchmark: driver-style line, default line (+ stage breakdown), power/clock trace over a long region
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver.json" 2> "$OUT/${TAG}_bench_driver.stderr"
python bench.py --stage-profile "$OUT/${TAG}_stages.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.stderr"
python bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-extras --no-profile --smi-trace "$OUT/${TAG}_power_clock_trace.json" > "$OUT/${TAG}_bench_long.json" 2>/dev/null
python bench.py --mode ir_se --no-cpu-baseline > "$OUT/${TAG}_bench_ir_se.json" 2>/dev/null
python bench.py --frame 1920x1080 --no-cpu-baseline > "$OUT/${TAG}_bench_1080p.json" 2>/dev/null
python bench.py --batch 1 --gallery 10000 --no-cpu-baseline > "$OUT/${TAG}_bench_config1.json" 2>/dev/null
python bench.py --batch 1 --gallery 10000 --no-cpu-baseline --fp32 > "$OUT/${TAG}_bench_config1_fp32.json" 2>/dev/null   # BASELINE configs[1] as labelled: fp32 recogniser
python bench.py --exact-match --no-cpu-baseline --no-extras > "$OUT/${TAG}_bench_exact_match.json" 2>/dev/null         # primary region with the exact fp32 scan (SURVEY 8(d)'s match)
python bench.py --faces 1 --no-cpu-baseline > "$OUT/${TAG}_bench_k1.json" 2>/dev/null
FRT_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | head -1 > "$OUT/${TAG}_bench_rccl_1rank.json"
FRT_BENCH_FORCE_DIST=1 python bench.py --sharded-gallery --batch 64 --gallery 1250000 --no-cpu-baseline 2>/dev/null | head -1 > "$OUT/${TAG}_bench_sharded_1rank.json"
# 3b. the drop-in shells measured (src/app.cpp:304-310 through include/frt/*.h, 1 and 8 threads, N = 1M)
python tools/dropin_bench.py --gallery 1000000 --threads 1 8 --iters 150 --out "$OUT/${TAG}_dropin_bench.json" > /dev/null 2>&1
# 3b'. round 4: the same call sequence from 8 / 32 request threads on ONE detector + ONE recogniser with request coalescing (frt_coalescer_*)
for t in 8 32; do python tools/dropin_bench.py --gallery 1000000 --threads $t --iters 150 --coalesce 32 --window-us 100 --out "$OUT/${TAG}_dropin_coalesce_t$t.json" > /dev/null 2>&1; done
python tools/dropin_bench.py --gallery 1000000 --threads 8 --iters 100 --shared --out "$OUT/${TAG}_dropin_shared_plain_t8.json" > /dev/null 2>&1
# 3b''. round 4: one process / every visible device (tests/cpp/multi_device_pipeline.cpp); BASELINE configs[3] per-rank shape (4 frames per step)
python tools/multi_device_bench.py --steps 100 --out "$OUT/${TAG}_multi_device.json" > /dev/null 2>&1
python bench.py --batch 4 --no-cpu-baseline --steps 300 --resident > "$OUT/${TAG}_bench_b4_resident.json" 2>/dev/null
python bench.py --batch 4 --no-cpu-baseline --steps 300 > "$OUT/${TAG}_bench_b4.json" 2>/dev/null
# round 5: consecutive calls sharing a recogniser pass (frt_pipeline_set_pairing): pairs and groups of four, 4 frames per call and K = 1
for g in 2 4; do python bench.py --batch 4 --no-cpu-baseline --steps 300 --no-extras --pair $g > "$OUT/${TAG}_bench_b4_pair$g.json" 2>/dev/null; done
python bench.py --faces 1 --no-cpu-baseline --no-extras --pair 4 > "$OUT/${TAG}_bench_k1_pair4.json" 2>/dev/null
# 3c. small batches: the small-batch recogniser path against the strip kernels and the oracle; per-launch table of a 1- and a 4-face pass
python tools/small_batch_parity.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/${TAG}_small_batch_parity.txt"
for nf in 1 4 16 32; do NF=$nf bash tools/layer_table.sh "A=1"; done > "$OUT/${TAG}_small_batch_layers.txt" 2>&1
python tools/ks_check.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/${TAG}_medium_batch_parity.txt"
python tools/prof_match_small.py 2>/dev/null | grep queries > "$OUT/${TAG}_match_small.txt"
# 4. the microbenchmarks DESIGN.md quotes (sources in tools/ubench/*.hip)
for P in clock_probe occ_probe mfma_f32_order mfma_pk_overlap; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w -o /tmp/$P "$ROOT/tools/ubench/$P.hip" 2>/dev/null && timeout 120 /tmp/$P > "$OUT/${TAG}_$P.txt" 2>&1
done
# 5. round 4: the detector alone (kernel trace of tools/prof_det.py at 32 and 4 frames), dwpw_wave_kernel against dwpw_mfma_kernel bit for bit
#    (tuning build), and the stand-alone harness of the wave kernel on its three shapes
cd /tmp
for B in 32 4 1; do
  rm -rf /tmp/prof_det && mkdir -p /tmp/prof_det
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o st -- python "$ROOT/tools/prof_det.py" $B 5 > /dev/null 2>&1
  cp "$(find /tmp/prof_det -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_det_kernel_stats_b$B.csv" 2>/dev/null
done
cd "$ROOT"
python tools/dwpw_wave_check.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/${TAG}_dwpw_wave_check.txt"
FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so FRT_DET_STEM_CHECK=1 python tools/stem_check_run.py 2>&1 | grep "stem check" > "$OUT/${TAG}_stem_check.txt"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFRT_TUNING -w -Iface-recognition-cpp-tensorrt_amd/csrc -o /tmp/dwpw_wave_bench tools/ubench/dwpw_wave_bench.hip 2>/dev/null &&
  for A in "32" "2" "32 64 80" "2 64 80" "32 256 20" "2 256 20"; do FRT_DWPW_WAVE_ANYB=1 timeout 60 /tmp/dwpw_wave_bench $A; done > "$OUT/${TAG}_dwpw_wave_bench.txt" 2>&1
# 6. round 5: detector PMC passes (waves, cycles, waits, instruction mix, HBM bytes) over the detector alone; probes and harnesses
{
  echo "rocprofv3 --pmc (separate passes, no trace domains) over tools/prof_det.py 32 2 (detector alone, 32 frames of 640x640), medians per launch, summed over the chip."
  echo "FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE NOT doubled here).  SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles."
  for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
    TARGET='prof_det.py 32 2' KFILTER='det_stem|dwpw_wave_kernel<128|conv3x3_split|pw_mfma|dwpw_mfma_kernel<2, 2, 32, 1, 2|dwpw_row4|decode|nms|heads' bash tools/pmc_pass.sh "$G"
  done
} > "$OUT/${TAG}_det_pmc.txt" 2>&1
[ -x tools/ubench/lds_dma_raw ] && timeout 120 tools/ubench/lds_dma_raw > "$OUT/${TAG}_lds_dma_raw.txt" 2>&1
[ -x tools/ubench/det_conv3h_bench ] && { timeout 300 tools/ubench/det_conv3h_bench 32; timeout 120 tools/ubench/det_conv3h_bench 4; timeout 120 tools/ubench/det_conv3h_bench 1; } > "$OUT/${TAG}_det_conv3h_bench.txt" 2>&1
[ -x tools/ubench/plane_stride ] && timeout 300 tools/ubench/plane_stride > "$OUT/${TAG}_plane_stride.txt" 2>&1
# 7. what each stage costs the pipelined step (tuning build, FRT_PIPE_ABLATE bit 0 = no detector network, 1 = no recogniser network, 2 = no match)
{
  echo "bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-extras, tuning build, FRT_PIPE_ABLATE (bit 0: no detector network after the first calls, bit 1: no"
  echo "recogniser network, bit 2: no match); faces/s and ms per step; K = 4 (the metric's configuration) and K = 1"
  for A in 0 1 2 4 5 6; do
    for K in 4 1; do
      printf "ablate %d  K %d  " $A $K
      FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so FRT_PIPE_ABLATE=$A python bench.py --steps 60 --warmup 12 --faces $K --no-cpu-baseline --no-extras 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
    done
  done
} > "$OUT/${TAG}_stage_ablation_raw.txt" 2>&1
for f in "$OUT"/${TAG}_det_kernel_stats_b*.csv; do echo "== $(basename $f)"; python tools/det_table.py "$f"; done > "$OUT/${TAG}_det_tables.txt" 2>&1
# 8. round 6: the adaptive mode at 2 - 11 calls in flight (4-frame calls), the fp32 recogniser pass per layer, the two fp32-MFMA microbenchmarks
python tools/proxy_only.py 2 3 4 5 6 8 11 2>&1 | grep depth > "$OUT/${TAG}_adaptive_depth_sweep.txt"
{ FRT_PROF_FP32=1 bash tools/trace_layers.sh 4 4; FRT_PROF_FP32=1 bash tools/trace_layers.sh 8 4 | tail -1; } > "$OUT/${TAG}_fp32_layers.txt" 2>&1
for P in mfma_f32_chain mfma_4x4_fma; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w -o /tmp/$P "$ROOT/tools/ubench/$P.hip" 2>/dev/null && timeout 120 /tmp/$P > "$OUT/${TAG}_$P.txt" 2>&1
done
python -c "import __graft_entry__ as e; e.smoke()" > "$OUT/${TAG}_smoke.txt" 2>&1
ls -la "$OUT"
