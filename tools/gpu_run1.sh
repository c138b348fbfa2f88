cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_vae.py tests/test_pipeline.py tests/test_inference_script.py -m gpu -q -x 2>&1 | tail -4
