mkdir -p gpurun_out/r03p
F="--steps 2 --warmup 1 --settle-steps 1 --blocks-per-step 16 --no-cpu-baseline --no-other-configs --no-dropin"
python bench.py $F > gpurun_out/r03p/a.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py $F > gpurun_out/r03p/b.json 2>/dev/null
GPU_MAX_HW_QUEUES=2 python bench.py $F > gpurun_out/r03p/c.json 2>/dev/null
for f in a b c; do grep -o '"ms_per_batch": [0-9.]*, "ms_per_batch_single_stream": [0-9.]*' gpurun_out/r03p/$f.json | head -2; done
