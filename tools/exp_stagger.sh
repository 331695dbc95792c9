set -e
cd $GRAFT_REPO_ROOT
for s in 0 13000 6500; do for c in 4 2 8; do
  if [ $s = 0 ] && [ $c != 4 ]; then continue; fi
  echo "=== stagger $s classes $c"; OVHIP_GEMM_STAGGER=$s OVHIP_GEMM_STAGGER_CLASSES=$c SLOTS=20 python tools/gemm_stamps.py 2>&1 | grep -E "^fc|^qkv|^out|^proj" 
done; done
