# Everything profiles/r03/ holds, collected on one box (gpurun_out/ is scratch; copy what is judged into profiles/r03/).
# usage: bash tools/profile_r03.sh tests | rest   (two gpurun calls: each stays under the 1200 s limit)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
if [ "$1" = tests ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
grep -E "passed|failed" $O/pytest.log | tail -1
grep "ref shapes\|bs16" $O/pytest.log > $O/reference_shapes.log || true
timeout -k 10 600 python bench.py --stages > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-400
exit 0
fi
timeout -k 10 200 python tools/second_bench.py > $O/second.log 2>&1
timeout -k 10 200 python tools/sorted_gemm_bench.py > $O/spconv_gemm_layers.log 2>&1
timeout -k 10 200 python tools/spconv_fwd_bench.py > $O/spconv_forward.log 2>&1 || true
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-full-rewrite > $O/bench_under_rocprof.json 2> $O/tr_bench.err
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/tools/spconv_trace.py > $O/pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/tools/spconv_trace.py > $O/pmc_b.log 2>&1
cd $R
for x in a b; do python tools/pmc_summary.py $O/pmc_$x sc_ 0 > $O/spconv_gemm_pmc_$x.json; rm -rf $O/pmc_$x; done
python tools/ktrace_last.py $O/tr_bench 110 vxl_keybin > $O/bench_step_timeline.txt || true
cp $(find $O/tr_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/tr_bench
bash tools/vx_pmc_collect.sh > $O/vx_pmc.log 2>&1 || true
cp gpurun_out/voxelize_pmc.json $O/voxelize_pmc.json
head -8 $O/bench_kernel_stats.csv | cut -c1-200
