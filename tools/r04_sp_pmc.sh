# PMC passes for the sparse implicit GEMM: r03's passes c / d / e asked for 5 TA / TCP / TCC counters each and
# died with "error code 38: Request exceeds the capabilities of the hardware to collect" (gfx950: TCC 4 slots per pass; TA / TCP
# 2-3).  Here: <= 3 TA / TCP and <= 3 TCC counters per pass, a short timeout per pass (an aborted rocprofv3 leaves the child idle),
# each pass once, and a pass that fails does not stop the others.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_sp_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  timeout -k 5 110 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -- python3 $R/tools/spconv_trace.py > $O/pmc_$1.log 2>&1
  echo "pass $1 rc=$? $(grep -c . $O/pmc_$1.log) log lines; $(grep -m1 -i 'error code\|exceeds' $O/pmc_$1.log)"
  (cd $R && python tools/pmc_summary.py $O/pmc_$1 sc_implicit 0 > $O/pmc_$1.json 2>/dev/null)
  rm -rf $O/pmc_$1
}
run ta "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
run tcp1 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
run tcp2 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
run tcp3 "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run tcc2 "TCC_BUSY_avr TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
run sq "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS"
cd $R
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_sp_pmc/pmc_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    for k,v in d.items():
        if "<2, 16, 1>" in k or "<1, 8" in k: print(f.split("/")[-1], k[:60], v)
P
