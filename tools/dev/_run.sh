cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/dev/trace_clip.py 24 | head -5
python tools/dev/stage_line.py 2>&1 | tail -1
python tools/dev/stage_line.py 2>&1 | tail -1
