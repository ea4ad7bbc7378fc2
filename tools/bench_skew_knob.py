#!/usr/bin/env python3
"""The two power-law legs of bench.py (MovieLens-10M shape, Zipf-0.8 items over cfg 2) alone: `python tools/bench_skew_knob.py ml|zipf`
(under MFX_K1_BURST_RUN etc.: tools/r05_burst_knobs.sh)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mfrec_amd import _shim as S, synth
S.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "ml"
k = 64
if which == "ml":
    m = bench.also_k1_ml10m(k)
else:
    nu, ni, nnz = 100000, 50000, 10000000
    m = bench.also_k1_skew(nu, ni, nnz, k, synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), float(os.environ.get("ZIPF", "0.8")))
print("%s: %.2f ms  %.1f M/s  %s" % (which, m["device_ms_per_step"], m["value"] / 1e6, m["roofline"]["kernel"][:60]))
