R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_pmcvx; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/vx_bench.py --algos 3 --iters 10 --resident > $O/p$i.log 2>&1
  python3 $R/tools/pmc_summary.py $O/p$i vxl_ 5 > $O/p$i.json
  rm -rf $O/p$i
done
cat $O/p*.json
