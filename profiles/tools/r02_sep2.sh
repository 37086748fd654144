#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02sep; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -s -k "sharded_forms" > $O/pytest_forms.log 2>&1; rc=$?; echo "forms rc=$rc"; grep -E "worst rel|passed|failed|^E " $O/pytest_forms.log | cut -c1-250 | tail -30
[ $rc = 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_multirank_hip.py -x -q -s > $O/pytest.log 2>&1; rc=$?; echo "multirank rc=$rc"; grep -E "separable|m-vector|passed|failed|^E " $O/pytest.log | cut -c1-250 | tail -30
