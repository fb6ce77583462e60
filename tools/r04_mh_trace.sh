R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_mh -- python3 $R/tools/config_trace.py multihead > $O/mh_trace.log 2>&1
cd $R
cp $(find $O/tr_mh -name "*kernel_stats.csv" | head -1) $O/multihead_kernel_stats.csv
rm -rf $O/tr_mh
head -40 $O/multihead_kernel_stats.csv | cut -c1-160
