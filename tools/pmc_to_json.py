#!/usr/bin/env python3
"""Turn the passes of tools/pmc_round.sh <tag> (merged back under gpurun_out/) into profiles/: the per-kernel CSV rows of
our kernels as profiles/<prefix>_pmc_*.csv and profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic).
usage: pmc_to_json.py <tag> <profiles-prefix, e.g. r03_a>"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]


def rows(name):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{name}", "**", "*counter_collection.csv"), recursive=True)
    out = []
    for f in fs:
        with open(f) as fh:
            out += [r for r in csv.DictReader(fh) if "pde::" in r["Kernel_Name"] or "adi_bwd_asm" in r["Kernel_Name"]]
    return out


def mean(rs, kernel, counter):
    v = [float(r["Counter_Value"]) for r in rs if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v) if v else None


def dump(name, rs):
    if not rs:
        return
    with open(os.path.join(ROOT, "profiles", f"{prefix}_pmc_{name}.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rs[0].keys()))
        w.writeheader()
        w.writerows(rs)


P = {n: rows(n) for n in ("fetch", "write", "sq1", "sq2", "tiny_fetch", "tiny_write", "secondary_fetch", "secondary_write",
                          "cfg4_fetch", "cfg4_write")}
# the backward of the headline schedule runs as the assembly kernel since round 4 (the HIP kernel's masked body behind it)
BWD = "adi_bwd_asm" if any("adi_bwd_asm" in r["Kernel_Name"] for r in P["fetch"]) else "adi_bwd_kernel"
for n, rs in P.items():
    dump(n, rs)
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
j = {"commit": commit,
     "source": f"profiles/{prefix}_pmc_fetch.csv + {prefix}_pmc_write.csv (separate --pmc passes of `bench.py --steps 3 --warmup 1`, "
               f"tools/pmc_round.sh); explicit kernels: {prefix}_pmc_tiny_*.csv (tools/prof_tiny.py, cfg5 256x64x64x64)",
     "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md "
                   "HBM section); WRITE_SIZE as read; both are KB per dispatch"}
j["bwd_kernel"] = BWD
for k, key in ((BWD, "adi_bwd_kernel"), ("adi_fwd_kernel", "adi_fwd_kernel")):
    f, w = mean(P["fetch"], k, "FETCH_SIZE"), mean(P["write"], k, "WRITE_SIZE")
    if f is not None and w is not None:
        j[key + "_bytes_per_launch"] = int((2 * f + w) * 1024)
        j[key + "_fetch_size_kb"], j[key + "_write_size_kb"] = f, w


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("pde::", "")
    return name.split("(")[0][:80]


def per_step(work, steps):
    """HBM bytes of ALL this library's launches of one forward+backward of a workload, and the per-kernel split"""
    fr, wr = P[work + "_fetch"], P[work + "_write"]
    if not fr or not wr:
        return
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in fr:
        if r["Counter_Name"] == "FETCH_SIZE":
            t = tot[short(r["Kernel_Name"])]
            t[0] += float(r["Counter_Value"]); t[2] += 1
    for r in wr:
        if r["Counter_Name"] == "WRITE_SIZE":
            tot[short(r["Kernel_Name"])][1] += float(r["Counter_Value"])
    j[work + "_bytes_per_step"] = int(sum((2 * f + w) for f, w, _ in tot.values()) * 1024 / steps)
    j[work + "_kernels"] = {k: {"launches_per_step": n / steps, "mb_per_launch": round((2 * f + w) * 1024 / n / 1e6, 2)}
                            for k, (f, w, n) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:12]}


per_step("secondary", 6)        # tools/prof_secondary.py runs six steps
per_step("cfg4", 3)             # tools/prof_cfg4.py three
for k in ("explicit5_fwd_wave", "explicit5_bwd_wave"):
    f, w = mean(P["tiny_fetch"], k, "FETCH_SIZE"), mean(P["tiny_write"], k, "WRITE_SIZE")
    if f is not None and w is not None:
        j[k + "_bytes_per_launch"] = int((2 * f + w) * 1024)
        j[k + "_fetch_size_kb"], j[k + "_write_size_kb"] = f, w
valu = {key: mean(P["sq1"], k, "SQ_INSTS_VALU") for k, key in ((BWD, "adi_bwd_kernel"), ("adi_fwd_kernel", "adi_fwd_kernel"))}
if all(v is not None for v in valu.values()):
    valu["source"] = f"profiles/{prefix}_pmc_sq1.csv"
    j["sq_insts_valu_per_launch"] = valu
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fh:
    json.dump(j, fh, indent=1)
print(json.dumps(j, indent=1))
