set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
timeout -k 10 280 python bench.py --stages > $O/bench.log 2>&1
timeout -k 10 200 python tools/second_bench.py > $O/second.log 2>&1
timeout -k 10 200 python tools/spconv_fwd_bench.py > $O/spconv_forward.log 2>&1
timeout -k 10 100 python tools/spconv_host_time.py >> $O/spconv_forward.log 2>&1
timeout -k 10 200 python tools/sorted_gemm_bench.py > $O/spconv_gemm_layers.log 2>&1
timeout -k 10 200 python tools/spconv_gemm_share.py > $O/spconv_gemm_share.log 2>&1
timeout -k 10 300 python tools/config_stage_bench.py > $O/configs.log 2>&1
timeout -k 10 200 python tools/pvrcnn_ops_bench.py > $O/pvrcnn_ops.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_sp -- python3 $R/tools/spconv_trace.py > $O/tr_sp.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_pv -- python3 $R/tools/config_trace.py pvrcnn > $O/tr_pv.log 2>&1
cd $R
python tools/ktrace_last.py $O/tr_sp 110 > $O/spconv_forward_timeline.txt
python tools/ktrace_last.py $O/tr_pv 110 fps_bucket_kernel > $O/pvrcnn_forward_timeline.txt
tail -1 $O/bench.log | cut -c1-300
