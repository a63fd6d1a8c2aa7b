# round 5: PMC passes (separate --pmc runs per MI355X_MICROARCH.md) over the launch mix of the dominant kernel family (bias-free dX GEMMs)
# and of the lowest-fraction one (fc1 + GELU) -> gpurun_out/r05_{dx,fc1}_pmc.json (copied to profiles/, read by bench.py's roofline.traffic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TREE=${TREE:-unknown}
mkdir -p $R/gpurun_out
for fam in dx fc1; do
  args=""
  for pass in "FETCH_SIZE:fetch" "WRITE_SIZE:write" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE:mfma" "TCC_HIT_sum TCC_MISS_sum:l2"; do
    ctr=${pass%%:*}; tag=${pass##*:}
    d=$R/gpurun_out/pmc_r5_${fam}_$tag
    rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python3 $R/tools/family_one.py $fam > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    f=$(ls $d/*counter_collection.csv $d/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] || { echo "no counter csv for $fam $tag"; exit 1; }
    cp $f $R/gpurun_out/pmc_r5_${fam}_$tag.csv
    args="$args $tag=$R/gpurun_out/pmc_r5_${fam}_$tag.csv"
    rm -rf $d
  done
  python3 $R/tools/pmc_family_json.py $fam gemm_nt_pers_kernel $TREE $R/gpurun_out/r05_${fam}_pmc.json $args
done
