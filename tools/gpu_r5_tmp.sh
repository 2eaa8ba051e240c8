#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/stage_bench.py small 50,51,52,53,54,55 2>&1 | grep -i "upd_a\|app_a"
timeout 600 python tools/stage_bench.py medium 50,51,52,53,54,55 2>&1 | grep -i "upd_a\|app_a"
