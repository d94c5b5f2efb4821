mkdir -p gpurun_out/r04r
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04r/t2.txt
timeout 300 python tools/latency.py 2>&1 | tail -6 > gpurun_out/r04r/lat.txt
timeout 600 python bench.py --config pq --steps 20 --warmup 3 > gpurun_out/r04r/pq.json 2> gpurun_out/r04r/pq.err
