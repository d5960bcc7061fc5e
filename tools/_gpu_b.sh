python -m pytest tests/test_gpu_train_step.py -x -q > gpurun_out/r04r_train_tests.log 2>&1; echo rc=$? >> gpurun_out/r04r_train_tests.log
rm -f gpurun_out/r04r_train_bench.txt
for b in 64 128 256 384 512; do python tools/train_step_bench.py othello8 $b 1500 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04r_train_bench.txt; done
python tools/train_step_bench.py connect4 64 1500 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04r_train_bench.txt
tail -3 gpurun_out/r04r_train_tests.log; cut -c1-80 gpurun_out/r04r_train_bench.txt
