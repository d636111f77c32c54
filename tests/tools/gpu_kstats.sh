# per-kernel durations (rocprofv3 --kernel-trace --stats) of a config's step loop on several libraries:
#   gpu_kstats.sh <out-dir-name> "<cfg:weights> ..." lib1 lib2 ...   -> gpurun_out/<out>/kernel_stats_<cfg>_<weights>_<libtag>.csv
set -x
export TMPDIR=/tmp
R=$PWD; O=$PWD/gpurun_out/$1; CFGS=$2; shift 2
mkdir -p $O
cd /tmp
for cw in $CFGS; do
  for lib in "$@"; do
    tag=$(basename $lib .so); d=$O/prof_${cw%%:*}_${cw##*:}_$tag
    VMAPSTEP_LIBRARY=$R/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o hl -- python $R/bench.py --config ${cw%%:*} --weights ${cw##*:} --steps 200 --warmup 20 --timed-only --no-cpu-baseline > $d.log 2>&1 < /dev/null
    for f in $d/*/*kernel_stats.csv $d/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $O/kernel_stats_${cw%%:*}_${cw##*:}_$tag.csv && head -4 "$f" | cut -c1-150; done
    rm -rf $d
  done
done
