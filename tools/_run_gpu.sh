for S in 16 2; do FREDDY_GPU_SPARSE_ITEMS=$S python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2; done
python tools/soak_round3.py 30 2>&1 | tail -1
python bench.py --steps 100 --warmup 10 --cpu-sample 0 --no-recall --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d.get('timed_region_parity'), d['host_buffer_abi'])"
