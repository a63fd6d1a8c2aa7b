# round-5 PMC passes for the attention kernels of the product path (forward / two-phase backward; S = 133 with dropout: keep-bit words
# written by the forward and read by the backward), separate --pmc passes per MI355X_MICROARCH.md; sums over 3 dispatches of each kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "197 0.0" "133 0.1"; do
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE:mfma" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16:lds" "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${pass%%:*}; tag=${pass##*:}
  d=$R/gpurun_out/pmc_r5_attn_$tag
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python3 $R/tools/attn_one.py $shape > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  f=$(ls $d/*counter_collection.csv $d/*/*counter_collection.csv 2>/dev/null | head -1)
  for k in attn_fwd_kernel attn_bwd_kernel; do
    echo "== S p = $shape  $k  pass $tag (sums over 3 dispatches)"
    [ -n "$f" ] && python3 $R/tools/pmc_sum.py $f $k
  done
  rm -rf $d
done
done
