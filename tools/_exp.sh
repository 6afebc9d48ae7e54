cd /root/repo
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo exit=$? >> gpurun_out/gpu_tests.log
for r in 8 16 32 64; do
URGYM_RESET_ENVS=$r python bench.py --env UR5DynReach-v1 --num-envs 65536 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/v_r${r}_dyn64k.log 2>&1
URGYM_RESET_ENVS=$r python bench.py --env UR5ObsReach-v1 --num-envs 16384 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/v_r${r}_obs16k.log 2>&1
done
