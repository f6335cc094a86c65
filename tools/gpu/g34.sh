cd /root/repo
mkdir -p gpurun_out
python tools/bench_cab_conv.py 4 > gpurun_out/r04_g34.log 2>&1
python tools/bench_cab_conv.py 8 >> gpurun_out/r04_g34.log 2>&1
head -40 gpurun_out/bygrid_bench_swinir_medium.txt >> gpurun_out/r04_g34.log
