cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/dev/stage_line.py 2>&1 | tail -1
for v in x2 x8 k8 k4 cb3; do SLR_SFS_AMD_LIB=$PWD/slr-sfs_amd/lib/var_$v.so python tools/dev/stage_line.py 2>&1 | tail -1; done
done
