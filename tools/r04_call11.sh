R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_deconv.py tests/test_gpu_pointpillar_path.py -k "deconv or golden" -x -q > $O/deconv_test.log 2>&1 || { tail -40 $O/deconv_test.log; exit 1; }
tail -2 $O/deconv_test.log
for v in 1 0; do echo "== bench LIDAR_BEV_DECONV=$v"; LIDAR_BEV_DECONV=$v timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --no-full-rewrite --roofline-launches 5 --stages 2>&1 | grep -E "stages|value" | cut -c1-170; done
LIDAR_BEV_DECONV=1 timeout -k 10 200 python tools/second_bench.py 2>&1 | tail -2
LIDAR_BEV_DECONV=0 timeout -k 10 200 python tools/second_bench.py 2>&1 | tail -2
