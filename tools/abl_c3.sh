cd /tmp && export TMPDIR=/tmp
for A in 1 2 3; do
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/wg$A
  FRT_C3_WG_PER_CU=$A rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/wg$A -o det -- python $GRAFT_REPO_ROOT/tools/prof_det.py 32 2 > /dev/null 2>&1
done
