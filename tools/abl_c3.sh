cd /tmp && export TMPDIR=/tmp
for A in 1 2; do
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/dw$A
  FRT_DWPW_WG_PER_CU=$A rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/dw$A -o det -- python $GRAFT_REPO_ROOT/tools/prof_det.py 32 2 > /dev/null 2>&1
done
