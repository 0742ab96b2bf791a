"""bench.py's output contract and the consistency of its `roofline` object with the committed profiles."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_dominant_kernel_selection_matches_the_committed_summary():
    """The kernel bench.py times live is the largest single-launch-site row of the newest committed rocprofv3 summary, named exactly
    as rocprofv3 names it, and priced with the same bytes / FLOPs tools/kernel_report.py uses for that row."""
    from eae_amd import profile_hooks as ph
    stats = ph.newest_stats(512)
    assert stats and os.path.basename(stats).startswith("r02_"), stats
    name, row, skipped = ph.pick_dominant(stats)
    rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
    assert name in rows and name in ph.KERNELS
    # nothing with a launch site of its own is larger
    for r in rows.values():
        if r["Name"] in ph.KERNELS:
            assert float(r["TotalDurationNs"]) <= float(row["TotalDurationNs"])
    # rows skipped on the way down are helpers launched several times per step
    assert all(s["calls"] > int(row["Calls"]) for s in skipped), skipped
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_report as kr
    site, role, bpi, mf = ph.KERNELS[name]
    frag = name.split("(")[0].replace("void ", "")
    match = [t for t in kr.T if t[0] in name]
    assert match and match[0][2] == bpi and abs(match[0][3] - mf) < 1e-9, (frag, match)
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic_b512.json")))["kernels"]
    assert name in pmc and pmc[name]["traffic_bytes"] > 0 and 0.0 < pmc[name]["mfma_util"] < 1.0


@pytest.mark.gpu
def test_bench_line_contract_and_roofline_arithmetic():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-configs"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                     # ONE JSON line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "images/s" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-4
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) <= 0.01 * rf["achieved"]
    assert abs(rf["avg_launch_us"] - (rf["event_bracket_us"] - rf["empty_bracket_us"])) < 0.05
    prof = rf["from_committed_profile"]
    assert prof["summary"].startswith("profiles/r02_") and rf["traffic"] == prof["pmc_traffic_bytes_per_launch"]
    # the live duration and the committed rocprofv3 average of the same kernel agree (tracing changes how the streams line up)
    assert 0.6 <= rf["avg_launch_us"] / prof["rocprof_avg_us"] <= 1.4, (rf["avg_launch_us"], prof["rocprof_avg_us"])
