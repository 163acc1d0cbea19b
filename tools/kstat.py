"""Average durations of the channel-operator kernels from rocprofv3 --kernel-trace --stats CSV output directories:
    python tools/kstat.py <dir> [<dir> ...]   (optionally a substring to match as last argument after --)"""
import csv,glob,sys
for d in sys.argv[1:]:
    fs=glob.glob(d+"/**/*kernel_stats.csv",recursive=True)
    if not fs: print(d,"no stats"); continue
    for r in csv.DictReader(open(fs[0])):
        if "mix_" in r["Name"]: print(d.split("/")[-1], r["Name"][:64], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
