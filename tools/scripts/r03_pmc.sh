# round-3 PMC passes for the dominant kernel (fc1 + bias + GELU) -- separate --pmc passes, as MI355X_MICROARCH.md prescribes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE:mfma" "FETCH_SIZE:fetch" "WRITE_SIZE:write" "TCC_HIT_sum TCC_MISS_sum:l2"; do
  ctr=${pass%%:*}; tag=${pass##*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r3_$tag -o p -- python3 $R/tools/gemm_one.py fc1 fc1_dna tr_dna > $R/gpurun_out/pmc_r3_$tag.log 2>&1
  f=$(ls $R/gpurun_out/pmc_r3_$tag/*counter_collection.csv $R/gpurun_out/pmc_r3_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== pass $tag ($ctr): $f"
  [ -n "$f" ] && python3 $R/tools/pmc_sum.py $f gemm_nt_pp_kernel
done
