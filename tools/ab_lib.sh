#!/bin/bash
# Same-box A/B of two builds of libfrt.so: alternating legs of the benchmark's timed region (no side measurements).
#   tools/ab_lib.sh TAG OLD_LIB [legs] [-- bench args]      (OLD_LIB relative to the package directory, e.g. libfrt_r05.so)
# The old library is not in the repository (*.so is ignored): build it from the commit to compare against, e.g. round 5's final tree:
#   git worktree add /tmp/r05 17e97a2 && make -j16 -C /tmp/r05/face-recognition-cpp-tensorrt_amd/csrc && cp /tmp/r05/face-recognition-cpp-tensorrt_amd/libfrt.so face-recognition-cpp-tensorrt_amd/libfrt_r05.so
# bench.py --ab-old-lib loads it with the symbols it lacks skipped (warned on stderr); the product import of the binding is strict.
set -u
TAG=${1:-ab}; OLDLIB=${2:-libfrt_r05.so}; LEGS=${3:-3}
shift 3 2>/dev/null || true
[ "${1:-}" = "--" ] && shift
ARGS=${*:---steps 200}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
OLD="--ab-old-lib $ROOT/face-recognition-cpp-tensorrt_amd/$OLDLIB"
leg() {  # label, extra bench arguments...
  local label=$1; shift
  python bench.py $ARGS "$@" --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'])"
}
{
echo "# bench.py $ARGS  (old = $OLDLIB)"
for i in $(seq $LEGS); do
  leg old $OLD
  leg new
done
} > "$OUT/ab.txt" 2>&1
cat "$OUT/ab.txt"
