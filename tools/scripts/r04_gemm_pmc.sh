# round-4 re-run (the GEMM kernels are round 3's) of the round-3 PMC passes for the dominant kernel after the persistent rewrite (fc1 + bias + GELU: gemm_nt_pers_kernel<2, true, false, true>;
# the 768x768 transform stays on gemm_nt_pp_kernel) -- separate --pmc passes per MI355X_MICROARCH.md, one shape per run, 3 dispatches each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in fc1 fc1_dna; do
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE:mfma" "FETCH_SIZE:fetch" "WRITE_SIZE:write" "TCC_HIT_sum TCC_MISS_sum:l2"; do
  ctr=${pass%%:*}; tag=${pass##*:}
  d=$R/gpurun_out/pmc_r4q_${shape}_$tag
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python3 $R/tools/gemm_one.py $shape > $d.log 2>&1
  f=$(ls $d/*counter_collection.csv $d/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $shape pass $tag ($ctr), sums over 3 dispatches"
  [ -n "$f" ] && python3 $R/tools/pmc_sum.py $f gemm_nt_p
  rm -rf $d
done
done
