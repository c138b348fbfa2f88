cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/mfma_pmc
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/mfma_pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
python tools/pmc_by_grid.py 'gpurun_out/mfma_pmc/**/*counter_collection.csv' > gpurun_out/r03z_pmc_mfma_busy.txt
cat gpurun_out/r03z_pmc_mfma_busy.txt | cut -c1-260
