#!/bin/bash
# Diagnostic: kernel statistics of one workload script with the stock library and with variant builds (tools/variant.sh).
# usage: tools/prof_pair.sh <script.py> <tag> [tag ...]   -> gpurun_out/pair_<name>_{hip,<tag>}/ and a summary on stdout
export TMPDIR=/tmp
script=$1; shift
name=$(basename $script .py)
for lib in hip "$@"; do
  export PDECNN_LIB=$PWD/cnn-with-pde_amd/lib/libpdecnn_${lib}.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pair_${name}_${lib} -- python $script > gpurun_out/pair_${name}_${lib}.log 2>&1
  echo "== $name $lib"
  python - gpurun_out/pair_${name}_${lib} <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("   %-62s %5s %9.1f us" % (r["Name"].replace("pde::(anonymous namespace)::", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
