# final collection of round 3: PMC traffic of the current voxelize.hip first (bench.py reports `traffic` only for a matching sha),
# then the full GPU suite, the bench line and the in-step trace.  usage: bash tools/r03_final.sh a | b   (two gpurun calls)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
if [ "$1" = a ]; then
bash tools/vx_pmc_collect.sh > $O/vx_pmc.log 2>&1 || true
cp gpurun_out/voxelize_pmc.json $O/voxelize_pmc.json
cp gpurun_out/voxelize_pmc.json profiles/r03/voxelize_pmc.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
grep -E "passed|failed" $O/pytest.log | tail -1
grep "ref shapes\|bs16" $O/pytest.log > $O/reference_shapes.log || true
exit 0
fi
cp $O/voxelize_pmc.json profiles/r03/voxelize_pmc.json 2>/dev/null || true
timeout -k 10 600 python bench.py --stages > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-full-rewrite > $O/bench_under_rocprof.json 2> $O/tr_bench.err
cd $R
python tools/ktrace_last.py $O/tr_bench 110 vxl_keybin > $O/bench_step_timeline.txt || true
cp $(find $O/tr_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/tr_bench
grep -E "vxl_|tk_" $O/bench_kernel_stats.csv | cut -c1-120
