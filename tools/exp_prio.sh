cd $GRAFT_REPO_ROOT
for p in 0 1 0 1; do echo "=== epi_prio $p"; OVHIP_GEMM_EPI_PRIO=$p STAMPS=0 python tools/gemm_stamps.py 2>&1 | grep -E "avg of"; done
OVHIP_GEMM_EPI_PRIO=1 python tools/gemm_wave_stamps.py qkv fc 2>&1 | grep -v Warn
