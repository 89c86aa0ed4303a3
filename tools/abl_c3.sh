cd /tmp && export TMPDIR=/tmp
for A in 0 1 2 4 5 7; do
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/c64_$A
  FRT_C64_ABLATE=$A rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/c64_$A -o emb -- python $GRAFT_REPO_ROOT/tools/prof_embed.py 128 2 > /dev/null 2>&1
done
