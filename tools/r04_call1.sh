# round 4, GPU call 1: the whole GPU suite, the bench line, and the kernel trace of the CONTRACT voxeliser path in-step
# (roofline.frac reproducible from profiles/)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
grep -E "passed|failed" $O/pytest.log | tail -1
timeout -k 10 600 python bench.py --stages > $O/bench.log 2>&1 || { tail -30 $O/bench.log; exit 1; }
tail -1 $O/bench.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_contract -- python3 $R/bench.py --contract-only --steps 100 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_contract_under_rocprof.json 2> $O/tr_contract.err || { tail -20 $O/tr_contract.err; exit 1; }
cd $R
python tools/vx_trace_gap.py $O/tr_contract 6 141312000 > $O/voxelize_contract_trace.json
cp $(find $O/tr_contract -name "*kernel_stats.csv" | head -1) $O/bench_contract_kernel_stats.csv
rm -rf $O/tr_contract
grep -E "vxl_" $O/bench_contract_kernel_stats.csv | cut -c1-160
head -12 $O/voxelize_contract_trace.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04/bench_contract_under_rocprof.json").read().strip().splitlines()[-1])
r=d["roofline"]; print({k:r[k] for k in ("frac","ms_per_launch","kernel_sum_us","gap_us","first_kernel_us","last_kernel_us","bracket_ms_per_launch","bracket_armed_ms_per_launch","launches")})
P
