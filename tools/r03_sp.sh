R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_sp; mkdir -p $O
cd $R
timeout -k 10 200 python tools/sorted_gemm_bench.py > $O/spconv_gemm_layers.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/tools/spconv_trace.py > $O/pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/tools/spconv_trace.py > $O/pmc_b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_c -- python3 $R/tools/spconv_trace.py > $O/pmc_c.log 2>&1
cd $R
for x in a b c; do python tools/pmc_summary.py $O/pmc_$x sc_ 0 > $O/pmc_$x.json; rm -rf $O/pmc_$x; done
tail -25 $O/spconv_gemm_layers.log
