cd /root/repo
mkdir -p gpurun_out
bash tools/gpu/final.sh r04e
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final_r04e_driver_cmd.json
