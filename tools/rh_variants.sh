cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in ${RH_TAGS:-hip}; do
  PDECNN_LIB=$R/cnn-with-pde_amd/lib/libpdecnn_$t.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rh_$t -o x -- python $R/tools/perf_rh.py 128 > $R/gpurun_out/rh_$t.log 2>&1 || exit 1
  echo "== $t"; grep "fused=True" $R/gpurun_out/rh_$t.log
  python - $R/gpurun_out/rh_$t <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rh_" in r["Name"]: print("   ", r["Name"][32:70], r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
done
