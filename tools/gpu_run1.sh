cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "lora_merge" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_pipeline.py tests/test_inference_script.py -m gpu -q 2>&1 | tail -3
timeout 900 python tools/microbench.py gemm attn elem > gpurun_out/r03_microbench.log 2>&1
tail -60 gpurun_out/r03_microbench.log
