mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/z15_gpu_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/z15_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/z15_smoke.log 2>&1; rc=$?
tail -n 3 gpurun_out/z15_smoke.log
exit $rc
