cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
DMG_FLOW_TIMING=1 python tools/flow_timing.py 2>&1 | tail -19
python tools/config_lines.py 2>/dev/null | grep -E '"config": (2|5)' | cut -c1-200
