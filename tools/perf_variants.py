#!/usr/bin/env python3
"""Diagnostic: time the headline workload (cfg2 primary, 512x64x32x32, 10 Strang steps) with several builds of the
library (tools/variant.sh), one child process per build (PDECNN_LIB), and print forward / backward launch times
(HIP events inside the library) and the step time.  usage: perf_variants.py tag [tag ...]   ('hip' = the stock build)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag in sys.argv[1:]:
    lib = os.path.join(ROOT, "cnn-with-pde_amd", "lib", f"libpdecnn_{tag}.so")
    env = dict(os.environ, PDECNN_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                        "--no-secondary", "--no-configs"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{tag:10s} step {j['ms_per_step']:.4f} ms  fwd {j['roofline_fwd']['avg_launch_ms'] * 1e3:7.1f} us  "
              f"bwd {j['roofline']['avg_launch_ms'] * 1e3:7.1f} us  graph {j['hipgraph_replay']['ms_per_step']}", flush=True)
    except Exception as e:
        print(tag, "FAILED", repr(e), r.stdout[-300:], r.stderr[-600:], flush=True)
