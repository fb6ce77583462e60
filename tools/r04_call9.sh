R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_pointpillar_path.py -x -q > $O/pp_test.log 2>&1 || { tail -40 $O/pp_test.log; exit 1; }
tail -2 $O/pp_test.log
for sf in 1 0; do for sp in 2 1; do
echo "== SPARSE_FIRST=$sf SPLIT=$sp"; LIDAR_BEV_SPARSE_FIRST=$sf LIDAR_BEV_SPLIT=$sp timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --no-full-rewrite --roofline-launches 5 --stages 2>&1 | grep -E "stages|value" | cut -c1-170
done; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-full-rewrite --roofline-launches 5 > $O/bench_under_rocprof.json 2> $O/tr_bench.err
cd $R
python tools/ktrace_last.py $O/tr_bench 110 vxl_keybin > $O/bench_step_timeline.txt || true
cp $(find $O/tr_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/tr_bench
cat $O/bench_step_timeline.txt | cut -c1-150 | head -150
