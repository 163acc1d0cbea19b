for g in 4 8 16; do echo "PDE_G_FWD=$g"; PDE_G_FWD=$g python tools/perf_variants.py base; done
for g in 2 4 8; do echo "PDE_G_BWD=$g"; PDE_G_BWD=$g python tools/perf_variants.py base; done
