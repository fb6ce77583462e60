# PMC passes for the F(4x4) Winograd kernel (csrc/wino43_conv.hip): how busy the matrix pipe is and how much of the waves' time goes
# to VALU instructions — the counters behind DESIGN 3.12's issue model.  <= 3 SQ counters per pass, --kernel-trace only, separate passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_w43_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  WINO_BENCH_ONLY=1 timeout -k 5 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -- python3 $R/tools/wino_bench.py 16,128,124,108 16,64,248,216 > $O/pmc_$1.log 2>&1
  echo "pass $1 rc=$? $(grep -m1 -i 'error code\|exceeds' $O/pmc_$1.log)"
  (cd $R && python tools/pmc_summary.py $O/pmc_$1 wino_f43 3 > $O/pmc_$1.json 2>/dev/null)
  rm -rf $O/pmc_$1
}
run a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES"
run c "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
run d "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
cd $R
cat $O/pmc_*.json
