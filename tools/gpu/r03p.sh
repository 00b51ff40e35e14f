#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for t in 1 3 5 9; do timeout 900 python tools/qs_check.py 1500000 $t 300 2>&1 | tail -7; done
