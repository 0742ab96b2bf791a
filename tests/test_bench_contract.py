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
    as rocprofv3 names it; gate rows (one waiting wave) are never picked; the profile carries the commit it was taken at."""
    from eae_amd import profile_hooks as ph
    stats = ph.newest_stats("b512")
    assert stats and os.path.basename(stats).startswith(("r02_", "r03_", "r04_")), stats
    name, row, skipped = ph.pick_dominant(stats)
    rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
    assert name in rows and ph.site_of(name) is not None and "gate_kernel" not in name
    # nothing with a launch site of its own is larger
    for r in rows.values():
        if ph.site_of(r["Name"]) is not None:
            assert float(r["TotalDurationNs"]) <= float(row["TotalDurationNs"])
    # rows skipped on the way down are helpers launched several times per step
    assert all(s["calls"] > int(row["Calls"]) for s in skipped), skipped
    rnd = os.path.basename(stats)[:3]
    pmc = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_b512.json")))["kernels"]
    assert name in pmc and pmc[name]["traffic_bytes"] > 0 and 0.0 < pmc[name]["mfma_util"] < 1.0
    if rnd != "r02":
        assert ph._commit_of(stats) == json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_profile_meta.json")))["commit"]


def test_site_of_and_site_model_on_every_kernel_name_of_the_committed_summaries():
    """Every conv-layer kernel name in the committed summaries maps to the launch site of its layer and role (from its template
    arguments, whatever tile geometry), and the byte / FLOP model reproduces SURVEY.md 8d's hand-derived figures at 64x64."""
    from eae_amd import profile_hooks as ph
    S = ph.prof_site
    known = {"void igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 1, 0>(ConvArgs)": S(1, 0),
             "void igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 2, 1>(ConvArgs)": S(1, 1),
             "void igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 2, 1>(ConvArgs)": S(6, 1),
             "void igemm_s2_kernel<0, 128, 256, 64, 4, 4, 8, 2, 2>(ConvArgs)": S(4, 1),
             "void igemm2_s2_kernel<0, 128, 256, 64, 4, 4, 8, 1, 0, 1>(ConvArgs)": S(3, 0),
             "void igemm2_s2_kernel<1, 256, 128, 64, 4, 4, 4, 0, 0, 1>(ConvArgs)": S(4, 0),
             "void wgrad_s2_kernel<64, 32, 16, 8, 1, 2, 1>(WgradArgs)": S(1, 2),
             "void wgrad_s2_kernel<64, 32, 16, 8, 1, 1, 2>(WgradArgs)": S(6, 2),
             "void wgrad_s2_kernel<256, 128, 4, 4, 8, 0, 2>(WgradArgs)": S(4, 2),
             "void wgrad_s2_kernel<256, 128, 4, 4, 8, 2, 1>(WgradArgs)": S(3, 2),
             "void deconv4_loss_kernel<1>(Deconv4Args)": S(7, 0), "void edge_wgrad_kernel<0, 2>(EdgeWgradArgs)": S(0, 2),
             "void edge_wgrad_kernel<1, 1>(EdgeWgradArgs)": S(7, 2), "void edge_conv_kernel<1, 1>(EdgeArgs)": S(7, 1),
             "void edge_conv_kernel<0, 0>(EdgeArgs)": S(0, 0), "void igemm8_s2_kernel<1, 64, 32, 32, 16, 8, 1, 2, 1>(ConvArgs)": S(1, 1),
             "void wgrad_s2_kernel_g<64, 32, 16, 8, 1, 1, 2>(GroupPack<WgradArgs>, int)": S(6, 2),          # grouped twins (eae_group.h)
             "void igemm_s2_kernel_g<0, 32, 64, 64, 16, 8, 1, 1, 0>(GroupPack<ConvArgs>, int)": S(1, 0)}
    for n, s in known.items():
        assert ph.site_of(n) == s, n
    for n in ("adam_kernel(AdamArgs)", "void head_kernel<16>(HeadArgs)", "gate_kernel(GateArgs)", "void fc_nt_kernel<1, 0>(FcNtArgs)"):
        assert ph.site_of(n) is None
    X, Y1, Y2 = 3 * 64 * 64, 32 * 32 * 32, 16 * 16 * 64
    MF, EDGE = 2.0 * 256 * 288 * 64 / 1e6, 2.0 * 1024 * 27 * 32 / 1e6
    want = {S(1, 0): (2 * Y1 + 2 * Y2, MF), S(1, 1): (4 * Y2 + 4 * Y1, MF), S(6, 1): (4 * Y1 + 4 * Y2, MF), S(1, 2): (4 * Y2 + 2 * Y1, MF),
            S(6, 2): (2 * Y2 + 4 * Y1, MF), S(7, 0): (2 * Y1 + 4 * X + 8 * 64 * 64, EDGE), S(0, 2): (4 * X + 4 * Y1, EDGE),
            S(7, 1): (8 * 64 * 64 + 4 * Y1, EDGE), S(6, 0): (2 * Y2 + 2 * Y1, MF), S(3, 0): (2 * 8 * 8 * 128 + 2 * 4 * 4 * 256, MF)}
    for s, (b, mf) in want.items():
        role, bpi, m = ph.site_model(s, 64, 64)
        assert bpi == b and abs(m - mf) < 1e-9, (s, role, bpi, b)
    # 256x256 inputs: every map 16x larger
    for s in ph.all_sites():
        assert ph.site_model(s, 256, 256)[1] == 16 * ph.site_model(s, 64, 64)[1]
    assert len(ph.all_sites()) == 23
    for f in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r0[34]_bench_*_kernel_stats.csv"))):
        for r in csv.DictReader(open(f)):
            if any(k in r["Name"] for k in ("igemm", "wgrad_s2", "wgrad8", "edge_", "deconv4_loss")):
                assert ph.site_of(r["Name"]) is not None, (f, r["Name"])


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
    assert prof["summary"].startswith(("profiles/r02_", "profiles/r03_", "profiles/r04_")) and rf["traffic"] == prof["pmc_traffic_bytes_per_launch"]
    assert prof["summary_commit"]
    # the live duration and the committed rocprofv3 average of the same kernel agree (tracing changes how the streams line up)
    assert 0.6 <= rf["avg_launch_us"] / prof["rocprof_avg_us"] <= 1.4, (rf["avg_launch_us"], prof["rocprof_avg_us"])


@pytest.mark.gpu
def test_bench_side_workload_line_carries_its_own_roofline():
    """`bench.py --workload c2` (what tools/profile_round.sh profiles for BASELINE configs[1]): one JSON line, the metric's contract,
    the workload named, and a roofline object priced for THAT workload (its own committed summary, its own batch)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["unit"] == "images/s" and d["dtype"] == "bf16" and "configs[1]" in d["config"]["workload"]
    assert d["config"]["per_gpu_batch"] == 256 and abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-4
    from eae_amd import profile_hooks as ph
    role, bpi, mf = ph.site_model(rf["site"], 64, 64)
    assert rf["algorithmic_bytes_per_launch"] == 256 * bpi and ph.site_of(rf["kernel"]) == rf["site"]
    prof = rf["from_committed_profile"]
    assert prof["summary"].startswith(("profiles/r03_bench_c2_", "profiles/r04_bench_c2_")) and prof["summary_commit"]
    assert rf["traffic"] == prof["pmc_traffic_bytes_per_launch"]
