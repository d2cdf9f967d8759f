#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/pytest.log; tail -${TAIL:-40} gpurun_out/pytest.log
exit $rc
