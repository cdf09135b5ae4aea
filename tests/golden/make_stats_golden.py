#!/usr/bin/env python3
"""Generates tests/golden/stats_golden.json from the REFERENCE's own Stats class
(oracle/_ref/libfrog_refstats.so = /root/reference/registration/stats.cxx compiled
unmodified by oracle/Makefile).  Run in the build container only; the fixture
(inputs + expected outputs, no reference source) is what travels.

    make -C oracle ref && python tests/golden/make_stats_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle_api import Stats, ref_lib  # noqa: E402


def f32_list(a):
    # exact: every float32 is representable as a double, JSON carries 17 digits
    return [float(x) for x in np.asarray(a, np.float32)]


def main():
    if ref_lib() is None:
        raise SystemExit("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(20250919)
    cases = []

    # 1. EM fit + inlier probabilities + histogram on a planted two-component mixture
    n = 600
    sigma = np.where(np.arange(n) % 3 == 0, 60.0, 3.0)
    samples = (np.linalg.norm(rng.normal(size=(n, 3)), axis=1) * sigma).astype(np.float32)
    s = Stats("ref")
    s.add_slots(n); s.reset(); s.add_samples(samples); s.estimate()
    probes = [0.0, 0.05, 0.0999, 0.1, 0.10000000149011612, 0.5, 1, 2, 5, 8, 12, 20, 35, 50, 80, 200, 1000]
    cases.append({"name": "em_mixture", "max_size": 10000, "max_iterations": 10000, "epsilon": 1e-6,
                  "samples": f32_list(samples), "params": f32_list(s.params()),
                  "probe_d": probes, "probe_p": [float(np.float32(s.prob(d))) for d in probes],
                  "histogram": f32_list(s.histogram(1.0))})

    # 2. warm-started second fit (c1, c2, ratio carry over between refreshes)
    samples2 = (samples * np.float32(0.8)).astype(np.float32)
    s.reset(); s.add_samples(samples2); s.estimate()
    cases.append({"name": "em_warm_start", "start_params": cases[0]["params"], "samples": f32_list(samples2),
                  "params": f32_list(s.params())})

    # 3. iteration cap and epsilon floor
    s3 = Stats("ref", max_size=10000, max_iterations=3, epsilon=1e-6)
    s3.add_slots(n); s3.reset(); s3.add_samples(samples); s3.estimate()
    cases.append({"name": "em_three_iterations", "max_iterations": 3, "samples": f32_list(samples),
                  "params": f32_list(s3.params())})

    # 4. reservoir: which ordinals survive, over three refreshes (generator never reseeded)
    cap, virtual = 1000, 2500
    r = Stats("ref", max_size=cap)
    r.add_slots(virtual)
    kept = []
    for refresh in range(3):
        r.reset()
        r.add_samples(np.arange(virtual, dtype=np.float32))      # sample value = its ordinal
        kept.append([int(x) for x in r.samples()])
    cases.append({"name": "reservoir", "max_size": cap, "virtual_size": virtual, "kept_ordinals": kept})

    # 5. chipdf spot values
    xs = [0.0, 1e-3, 0.3, 1.0, 1.4142135, 3.0, 7.5, 13.0, 20.0]
    cases.append({"name": "chipdf", "x": xs, "y": [float(np.float32(ref_lib().refstats_chipdf(float(np.float32(x))))) for x in xs]})

    out = os.path.join(ROOT, "tests", "golden", "stats_golden.json")
    json.dump({"generator": "tests/golden/make_stats_golden.py",
               "source": "reference registration/stats.cxx + stats.h, compiled unmodified (oracle/Makefile target ref)",
               "cases": cases}, open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
