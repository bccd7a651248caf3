# One stream's 22-row prefill attention: splits per (kv head, unit) against kernel time, the gap behind it (write-back of the partial slabs) and the combine launch.
# usage (GPU box): bash profiles/prefill_splits_probe.sh      -- AB_ATTN_WGS bits 16.. = the prefill form's target workgroup count (isst_op_set_attn_tuning)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in 0 32 64 96 128 256; do
  export AB_ATTN_WGS=$((T << 16))
  D=$R/gpurun_out/ab_tmp; rm -rf $D
  echo "== prefill target workgroups $T (0 = default 512 -> 19 splits of one 64-slot span)"
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/profiles/lib_variant_step.py $R/infinisst_amd/libinfinisst_hip.so 1 12 2>/dev/null | grep "ms per step"
  S=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 $R/profiles/trace_gaps.py $S --window 0.3:0.9 | grep -E "llm_attn_prefill|llm_attn_combine|gemm_mid_kernel<2, 1, 7|^window"
  rm -rf $D
done
