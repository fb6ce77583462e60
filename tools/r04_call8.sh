set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest2.log 2>&1 || { tail -60 $O/pytest2.log; exit 1; }
tail -2 $O/pytest2.log
timeout -k 10 600 python bench.py --stages > $O/bench2.log 2>&1 || { tail -30 $O/bench2.log; exit 1; }
grep stages $O/bench2.log
tail -1 $O/bench2.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['extra']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_sum_us'])
print({k:e[k] for k in ('second_kitti','pvrcnn_kitti','second_multihead_nuscenes','spconv_gemm','h2d_inclusive_frames_per_s') if k in e})
"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-full-rewrite --roofline-launches 5 > $O/bench_under_rocprof.json 2> $O/tr_bench.err
cd $R
python tools/ktrace_last.py $O/tr_bench 110 vxl_keybin > $O/bench_step_timeline.txt || true
cp $(find $O/tr_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/tr_bench
head -25 $O/bench_kernel_stats.csv | cut -c1-150
