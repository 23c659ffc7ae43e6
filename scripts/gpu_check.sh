#!/bin/bash
# what the driver runs at round end, in one call: the whole GPU test suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -n 6 > gpurun_out/check_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/check_smoke.log
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; echo "bench exit $?" >> gpurun_out/check_smoke.log
cat gpurun_out/check_pytest.log gpurun_out/check_smoke.log; tail -c 600 gpurun_out/check_bench.json
