cd /root/repo
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo exit=$? >> gpurun_out/gpu_tests.log
python bench.py --env UR5DynReach-v1 --num-envs 65536 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/v_trim_dyn64k.log 2>&1
python bench.py --env UR5DynReach-v1 --num-envs 262144 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/v_trim_dyn256k.log 2>&1
python bench.py --env UR5ObsReach-v1 --num-envs 16384 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/v_trim_obs16k.log 2>&1
python bench.py --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --no-cpu-baseline --rollout > gpurun_out/v_trim_ori4k_rollout.log 2>&1
python bench.py --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --no-cpu-baseline --rollout --no-collision > gpurun_out/v_trim_ori4k_nocoll.log 2>&1
python bench.py --env UR5OriReach-v1 --num-envs 65536 --steps 300 --warmup 20 --no-cpu-baseline --rollout --no-collision > gpurun_out/v_trim_ori64k_nocoll.log 2>&1
