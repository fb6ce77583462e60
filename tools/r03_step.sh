set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_step; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_pointpillar_path.py tests/test_gpu_bench_paths.py tests/test_gpu_models_mirror.py tests/test_gpu_spconv.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 100 python tools/topk_bench.py 2>&1 | tail -2
timeout -k 10 300 python bench.py --stages --no-extra --no-cpu-baseline > $O/bench.log 2>&1; grep stages $O/bench.log; tail -1 $O/bench.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-full-rewrite > $O/under.json 2> $O/tr.err
cd $R
python tools/ktrace_last.py $O/tr 110 vxl_keybin > $O/timeline.txt || true
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/tr
grep -E "vxl_|tk_|anchor_|decode_|post_nms|nms_|pfn_|mbtopk|radixSort" $O/kernel_stats.csv | cut -c1-110
