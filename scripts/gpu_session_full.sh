#!/bin/bash
# GPU session: parity tests, bench, rocprof kernel trace.  Stops after any timeout/kill.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/pytest.log; tail -25 gpurun_out/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --extra > gpurun_out/bench.log 2>&1
rc=$?; echo "bench_exit=$rc" >> gpurun_out/bench.log; tail -5 gpurun_out/bench.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
rc=$?; echo "prof_exit=$rc" >> $GRAFT_REPO_ROOT/gpurun_out/prof.log; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof.log
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*stats*" | head
