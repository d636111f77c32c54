# round 2, call J: step_main_ws evidence - bench line of the background config, rocprofv3 kernel stats, hardware counters
set -x
mkdir -p gpurun_out/r2j/pmc
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r2j
timeout 200 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_background.json 2> $O/bench_background.err < /dev/null; tail -1 $O/bench_background.json | head -c 300; echo
timeout 120 python tests/tools/phase_profile.py background > $O/phases_background.txt 2>&1 < /dev/null
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config background --timed-only --steps 200 --warmup 20 > $O/prof_run.log 2>&1 < /dev/null
for C in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc/$tag -o p -- python $R/tests/tools/run_steps.py background 20 > $O/pmc/$tag.log 2>&1 < /dev/null
  echo "$tag rc=$?"
done
cd $R
for f in $O/prof/*kernel_stats.csv; do [ -f "$f" ] && head -4 "$f"; done
true
