cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -s -k "small_row_count or headline_size_backward" 2>&1 | grep -E "headline backward|passed|failed" | tail -5
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -6
