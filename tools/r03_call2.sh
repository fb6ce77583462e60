set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_c2; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 600 python bench.py --stages > $O/bench.log 2>&1 || { tail -20 $O/bench.log; exit 1; }
tail -1 $O/bench.log | cut -c1-1500
bash tools/vx_pmc_collect.sh > $O/vx_pmc.log 2>&1 || true
cp gpurun_out/voxelize_pmc.json $O/voxelize_pmc.json
