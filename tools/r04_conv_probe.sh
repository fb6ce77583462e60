# MIOpen solver survey for the stride-1 3x3 fp32 convolutions (one process per setting)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/conv_solver_probe.log; : > $L
run() { timeout -k 10 150 "$@" >> $L 2>&1 || echo "[failed/timeout] $*" >> $L; }
run python tools/conv_solver_probe.py nhwc 0 default
run python tools/conv_solver_probe.py nhwc 1 deterministic
run python tools/conv_solver_probe.py nchw 0 default
run python tools/conv_solver_probe.py nchw 1 deterministic
for s in ConvBinWinogradRxSf2x3g1 ConvBinWinogradRxSf3x2 ConvBinWinogradRxSf2x3 ConvBinWinogradRxS ConvWinoFuryRxS\<2-3\> ConvMPBidirectWinograd\<3-3\> ConvWinograd3x3MultipassWrW\<3-4\> ConvHipImplicitGemmFwdXdlops ConvHipImplicitGemm3DGroupFwdXdlops ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC ConvDirectNaiveConvFwd GemmFwdRest ConvCkIgemmFwdV6r1DlopsNchw ConvHipImplicitGemmGroupFwdXdlops; do
  for lay in nchw nhwc; do
    MIOPEN_DEBUG_FIND_ONLY_SOLVER="$s" run python tools/conv_solver_probe.py $lay 0 "only=$s"
  done
done
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0 run python tools/conv_solver_probe.py nhwc 0 no_asm_gtc_nhwc
grep -v "^$" $L | grep "^\[" | cut -c1-400
