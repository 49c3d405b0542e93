mkdir -p gpurun_out/r6m
MLOAM_SCENE_FAMILY=hard timeout -k 10 400 python scripts/soak_parity.py 200 711 2>&1 | tail -1 > gpurun_out/r6m/hard_parity.txt
MLOAM_SCENE_FAMILY=hard timeout -k 10 600 python scripts/soak_parity_frontend.py 60 712 track,segment,rough,voxel,select,odom_select 2>&1 | grep -v "ok trial" | tail -8 > gpurun_out/r6m/hard_frontend.txt
timeout -k 10 300 python scripts/soak_parity_frontend.py 100 713 select 2>&1 | tail -2 > gpurun_out/r6m/select.txt
cat gpurun_out/r6m/*.txt
