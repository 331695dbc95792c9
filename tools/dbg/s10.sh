set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > gpurun_out/s10_ops.log 2>&1 || { tail -30 gpurun_out/s10_ops.log; exit 1; }
tail -2 gpurun_out/s10_ops.log
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "tiny or batch_invariance or large14_224_features or hipgraph or zero_shot" > gpurun_out/s10_model.log 2>&1 || { tail -30 gpurun_out/s10_model.log; exit 1; }
tail -2 gpurun_out/s10_model.log
python tools/graph_probe.py > gpurun_out/s10_graph.log 2>&1; tail -12 gpurun_out/s10_graph.log
OVHIP_GEMM_SKINNY_TILES=0 python tools/graph_probe.py > gpurun_out/s10_graph_off.log 2>&1; tail -12 gpurun_out/s10_graph_off.log
