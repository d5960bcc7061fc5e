rm -f gpurun_out/r04v.txt
for pf in 1 2; do for b in 384 512; do AZ_TRAIN_PF=$pf python tools/train_step_bench.py othello8 $b 1500 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r04v.txt; done; done
cut -c1-90 gpurun_out/r04v.txt
