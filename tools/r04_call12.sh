R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for sp in 2 1; do echo "== SPLIT=$sp"; LIDAR_BEV_SPLIT=$sp timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --no-full-rewrite --roofline-launches 5 --stages 2>&1 | grep -E "stages|value" | cut -c1-170; done
cd /tmp && export TMPDIR=/tmp
LIDAR_BEV_SPLIT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-full-rewrite --roofline-launches 5 > $O/bench_under_rocprof_split1.json 2> $O/tr_bench.err
cd $R
python tools/ktrace_last.py $O/tr_bench 100 vxl_keybin > $O/bench_step_timeline_split1.txt || true
rm -rf $O/tr_bench
cat $O/bench_step_timeline_split1.txt | cut -c1-140
