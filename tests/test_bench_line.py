"""The committed bench lines are self-contained: every fraction in them can be recomputed from the line's own fields
(what a reader of profiles/ does), and the contract's keys are there."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]_a_bench_n1*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06_bench_n1*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_fractions_follow_from_the_lines_own_fields(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["config"]["workload"] and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "ta+valu+lds") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r.get("frac_rule")
    steady, build = r["steady_launches"], r.get("list_writing_launches", {"launches": 0, "avg_launch_ms": 0.0})
    walked = steady["launches"] * (20.0 * r["walked_half_links_per_steady_launch"] + 12.0 * r["points_owned"]) \
        + build["launches"] * (20.0 * r["half_links_owned"] + 12.0 * r["points_owned"])
    ms = steady["launches"] * steady["avg_launch_ms"] + build["launches"] * build["avg_launch_ms"]
    achieved = walked / (ms * 1e-3) / 1e9
    assert abs(achieved - r["achieved"]) < 1e-6 * achieved
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # frac prices what the launches walked and cannot exceed the peak; frac_algorithmic_equiv prices every owned half-link,
    # walked or not -- a saving of work, not a bandwidth -- and MAY exceed 1 (a list that leaves out 30 % of the links does)
    assert r["frac"] <= 1.0
    assert r["frac"] <= r["frac_algorithmic_equiv"] or r["walked_half_links_per_steady_launch"] == r["half_links_owned"]
    equiv = (steady["launches"] + build["launches"]) * r["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e9 / r["peak"]
    assert abs(equiv - r["frac_algorithmic_equiv"]) < 1e-6 * equiv
    if "iteration" in d:
        it = d["iteration"]
        sched = d["config"]["schedule"]
        L, P = r["half_links_owned"], r["points_owned"]
        images = int(d["config"]["workload"].split(":")[1].split("images")[0])
        total = sched["linear"] * (20.0 * L + 36.0 * P)
        for la in it["lattices"]:
            g = la["dims"][0] * la["dims"][1] * la["dims"][2]
            total += la["iterations"] * (20.0 * L + 48.0 * P + 104.0 * images * g)
        assert abs(total - it["algorithmic_bytes"]) < 1e-9 * total
        assert abs(it["iteration_frac"] - total / it["elapsed_s"] / 8e12) < 1e-9
        assert it["iteration_frac_walked"] <= it["iteration_frac"] < 1.0
    if "cpu_baseline" in d:
        assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
