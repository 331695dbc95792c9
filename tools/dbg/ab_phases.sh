# A/B of two builds of libovhip.so on ONE device (run ON the GPU box): the in-tree library against openvision_amd/libovhip_alt.so
set -e
cd $GRAFT_REPO_ROOT
run() {
  python tools/gemm_stamps.py 2>&1 | grep -E "^(qkv|out|fc|proj): avg|K-tile us" 
  for i in 1 2; do python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['loss'])"; done
}
echo "== main"; run
cp openvision_amd/libovhip.so /tmp/main.so; cp openvision_amd/libovhip_alt.so openvision_amd/libovhip.so
echo "== alt"; run
cp /tmp/main.so openvision_amd/libovhip.so
echo "== main again"; run
