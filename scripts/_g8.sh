timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_real32.py tests/test_fortran.py -m gpu -x -q -k "not sharing_one_gpu and not configs3" 2>&1 | tail -4
LSQRHIP_SHARD_COPY_STREAMS=0 timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "several_ranks or overlapped_exchanges or loopback" 2>&1 | tail -3
timeout 600 python tests/fuzz_layouts.py 80 71 --engine 2>&1 | tail -3
LSQRHIP_SHARD_COPY=1 timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "single_process_sharded_handle" 2>&1 | tail -3
