cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "norm_modulate or race_probe" 2>&1 | tail -3
(echo "# tools/race_probe.py + tools/race_probe_attn.py + tools/determinism_probe.py on the last build of round 3"; timeout 600 python tools/race_probe.py 2>&1 | tail -25; timeout 600 python tools/race_probe_attn.py 2>&1 | tail -12; timeout 600 python tools/determinism_probe.py 2>&1 | tail -8) > gpurun_out/r03z_probes.txt
grep -v amdgpu.ids gpurun_out/r03z_probes.txt | tail -45
