cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_large_golden.py -m gpu -q -s -k native 2>&1 | grep -E "Error|assert|native 768|passed|failed" | head
python tools/dev/trace_clip.py 24 | head -5
