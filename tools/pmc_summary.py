"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                k = k.replace("void pde::(anonymous namespace)::", "").replace("pde::(anonymous namespace)::", "").split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "adi_" not in k:
                continue
            print(k, {c: sum(v) / len(v) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
