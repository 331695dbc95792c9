cd $GRAFT_REPO_ROOT
python tools/gemm_check.py 2>&1 | grep -v Warn
STAMPS=0 python tools/gemm_stamps.py 2>&1 | grep -E "avg of"
STAMPS=0 python tools/gemm_stamps.py 2>&1 | grep -E "avg of"
python tools/gemm_wave_stamps.py qkv fc 2>&1 | grep -v Warn
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or linear" 2>&1 | tail -3
