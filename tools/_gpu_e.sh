python -m pytest tests -m gpu -q -x > gpurun_out/r04f_tests.log 2>&1; echo rc=$? >> gpurun_out/r04f_tests.log
tail -6 gpurun_out/r04f_tests.log
python bench.py > gpurun_out/r04f_bench.log 2>&1; echo rc=$? >> gpurun_out/r04f_bench.log
tail -c 600 gpurun_out/r04f_bench.log
