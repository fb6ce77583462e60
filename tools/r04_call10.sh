R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pointpillar_path.py -k "wino or golden" -x -q > $O/wino_test.log 2>&1 || { tail -40 $O/wino_test.log; exit 1; }
tail -2 $O/wino_test.log
echo "== v4 (two waves per SIMD)"; WINO_BENCH_ONLY=1 timeout -k 10 300 python tools/wino_bench.py 2>&1 | grep wino
echo "== v3 (one wave per SIMD)"; LIDAR_WINO_V3=1 WINO_BENCH_ONLY=1 timeout -k 10 300 python tools/wino_bench.py 2>&1 | grep wino
echo "== v4 again"; WINO_BENCH_ONLY=1 timeout -k 10 300 python tools/wino_bench.py 2>&1 | grep wino
for v in 0 1; do echo "== bench LIDAR_WINO_V3=$v"; LIDAR_WINO_V3=$v timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --no-full-rewrite --roofline-launches 5 --stages 2>&1 | grep -E "stages|value" | cut -c1-170; done
