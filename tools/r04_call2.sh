set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pointpillar_path.py -k "wino or golden" -x -q > $O/wino_test.log 2>&1 || { tail -40 $O/wino_test.log; exit 1; }
tail -3 $O/wino_test.log
timeout -k 10 300 python tools/wino_bench.py > $O/wino_bench.log 2>&1 || { tail -20 $O/wino_bench.log; exit 1; }
cat $O/wino_bench.log
timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --stages > $O/bench_wino.log 2>&1 || { tail -20 $O/bench_wino.log; exit 1; }
grep stages $O/bench_wino.log; tail -1 $O/bench_wino.log | cut -c1-260
bash tools/r04_conv_probe.sh || true
bash tools/r04_sp_pmc.sh || true
