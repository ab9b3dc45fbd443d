cd $GRAFT_REPO_ROOT
timeout 200 python tools/timeline.py --opt gemv_ahead=0 2>&1 | grep -A6 "first-stage"
timeout 200 python tools/timeline.py --opt gemv_ahead=1 2>&1 | grep -A6 "first-stage"
timeout 200 python tools/timeline.py --opt gemv_ahead=1 2>&1 | grep -A8 "wo (gemv)"
