"""The evidence under profiles/ is self-consistent: every fraction a bench line quotes can be recomputed from the counter
files beside it, none exceeds 1, and the kernel-trace summary agrees with the duration the line measured with HIP events.
No GPU: this reads committed files only (round 1's review asked for exactly this property; the test keeps it true when
the profiles are re-installed)."""
import csv
import json
import os

import pytest

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
RND = "r3"          # the round whose files traffic.json and the bench lines were installed from (scripts/install_profiles.py)
KERNEL = "k_render<false, true, false, false>"
N_SIMD, N_CU, N_XCD = 1024, 256, 8
PIXELS = {"c2": 3840 * 2160, "c3": 3840 * 2160, "c4": 7680 * 4320 * 16}


def _load(name):
    with open(os.path.join(P, name)) as f:
        return json.load(f)


def _kernel(pmc):
    return [v for n, v in pmc.items() if KERNEL in n][0]


@pytest.mark.parametrize("w", ["c2", "c3", "c4"])
def test_traffic_json_follows_from_the_counter_files(w):
    t = _load("traffic.json")[w]
    k = _kernel(_load(f"{RND}_pmc_{w}.json"))
    m = lambda c: k[c]["mean"]
    cyc = m("GRBM_GUI_ACTIVE") / N_XCD
    assert t["kernel_cycles"] == pytest.approx(cyc)
    # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md), so reads are doubled; KB -> bytes
    assert t["k_render_hbm_bytes_per_launch"] == pytest.approx((2 * m("FETCH_SIZE") + m("WRITE_SIZE")) * 1024)
    assert t["valu_issue_frac"] == pytest.approx(m("SQ_INSTS_VALU") * 2.2 / N_SIMD / cyc, abs=6e-4)
    # the floor the bench line quotes as `frac`: every vector instruction at the guide's 2 clocks (MI355X_MICROARCH.md)
    assert t["valu_issue_frac_guide_clk"] == pytest.approx(m("SQ_INSTS_VALU") * 2.0 / N_SIMD / cyc, abs=6e-4)
    assert t["salu_issue_frac"] == pytest.approx(m("SQ_INSTS_SALU") / N_CU / cyc, abs=6e-4)
    assert t["vmem_return_frac"] == pytest.approx(m("SQ_INSTS_VMEM_RD") * 16.3 / N_CU / cyc, abs=6e-4)
    mix = _load(f"{RND}_valu_mix.json")["mean_issue_clocks_per_instruction"]
    assert t["valu_issue_frac_mix_raw"] == pytest.approx(m("SQ_INSTS_VALU") * mix / N_SIMD / cyc, abs=6e-4)
    assert t["valu_issue_frac_mix"] == pytest.approx(min(1.0, t["valu_issue_frac_mix_raw"]), abs=6e-4)
    for key in ("valu_issue_frac", "valu_issue_frac_guide_clk", "valu_issue_frac_mix", "salu_issue_frac", "vmem_return_frac"):
        assert 0.0 < t[key] <= 1.0, key


@pytest.mark.parametrize("w", ["c2", "c3", "c4"])
def test_bench_line_quotes_traffic_json_and_its_own_clock(w):
    t = _load("traffic.json")[w]
    d = _load(f"{RND}_bench_{w}.json")
    r = d["roofline"]
    assert d["config"]["workload"] == w and d["n_gpus"] == 1 and d["unit"] == "Mrays/s" and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(PIXELS[w] / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
    assert r["kernel_ms"] <= d["ms_per_step"]
    cands = r["candidates"]
    key_of = {"valu_issue": "valu_issue_frac_guide_clk"}
    bound_key = key_of.get(r["bound"], r["bound"] + "_frac")
    assert bound_key == max(cands, key=cands.get) and r["frac"] == cands[bound_key] <= 1.0
    assert r["frac"] == t[bound_key] and r["achieved"] / r["peak"] == pytest.approx(r["frac"])
    # every fraction in the object is <= 1, and what is not measured in the run says where it comes from
    assert all(v <= 1.0 for v in cands.values()) and r["hbm_frac"] <= 1.0
    fp = r["from_profiles"]
    assert fp["source_kind"].startswith("profiles (static)") and fp["source"].startswith("profiles/" + RND + "_pmc_")
    assert fp["valu_issue_frac_at_2p2_clk"] == t["valu_issue_frac"] <= 1.0
    assert "valu_issue_frac_mix" not in r and "valu_issue_frac_mix" not in fp         # round 2's estimate exceeded 1 on two workloads
    assert d["value_unconditional"] == pytest.approx(PIXELS[w] / (d["unordered_ms"] * 1e-3) / 1e6) and d["value_unconditional"] <= d["value"] * 1.02
    assert r["traffic"] == pytest.approx(t["k_render_hbm_bytes_per_launch"])
    assert r["hbm_frac"] == pytest.approx(r["traffic"] / (r["kernel_ms"] * 1e-3) / 8e12)
    requested = r["algorithmic_bytes_per_ray"] * PIXELS[w]
    assert r["reuse_factor"] == pytest.approx(requested / r["traffic"])
    assert r["requested_GBps"] == pytest.approx(requested / (r["kernel_ms"] * 1e-3) / 1e9)
    # 8 B per visited voxel + 40 B per triangle test + the pixel's outputs
    s = d["scene"]
    spp = d["config"]["spp"]
    assert r["algorithmic_bytes_per_ray"] == pytest.approx(8 * s["voxels_per_ray"] + 40 * s["tri_tests_per_ray"] + 20.0 / spp)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0


def test_kernel_trace_agrees_with_the_events():
    """rocprofv3 --kernel-trace --stats of the same command.  The traversal kernel runs in two settings in that process: the
    warm-up and timed steps (tile cover on) and the five launches with history of the HVRT_TILE_COVER=0 context behind them
    (`no_tile_cover_ms`; that context's first launch and the context without history run the give-up variant, a kernel of
    another name).  The summary's average over all of them is the mix of the two means the line measured with HIP events,
    less what the events also bracket: with the tile cover, the queue's two kernels and the wait for k_render_split in front
    of k_render (taken from the same summary), the gaps between those launches, and the end of k_render_split where it
    outlasts k_render."""
    d = _load(f"{RND}_bench_c2_under_rocprof.json")
    with open(os.path.join(P, f"{RND}_bench_c2_kernel_stats.csv")) as f:
        all_rows = list(csv.DictReader(f))
    rows = [r for r in all_rows if KERNEL in r["Name"]]
    assert len(rows) == 1
    calls, avg_ms, min_ms = int(rows[0]["Calls"]), float(rows[0]["AverageNs"]) * 1e-6, float(rows[0]["MinNs"]) * 1e-6
    n_cover, n_plain = d["steps"] + d["warmup"], 5
    assert calls == n_cover + n_plain
    in_front = 0.0
    for name in ("k_compact_count", "k_compact_queue", "k_wait_split_started"):
        r = [r for r in all_rows if name + "(" in r["Name"]]
        assert len(r) == 1, name
        in_front += float(r[0]["AverageNs"]) * 1e-6
    ev_ms = (n_cover * d["roofline"]["kernel_ms"] + n_plain * d["no_tile_cover_ms"]) / calls
    trace_ms = avg_ms + in_front * n_cover / calls
    assert trace_ms <= ev_ms <= trace_ms * 1.15, (trace_ms, ev_ms)
    assert min_ms <= d["roofline"]["kernel_ms"] <= min_ms * 1.40, (min_ms, d["roofline"]["kernel_ms"])


@pytest.mark.parametrize("w", ["c2", "c3", "c4"])
def test_the_line_without_the_tile_cover(w):
    """The same kernel with every tile walked (HVRT_TILE_COVER=0): its own counter passes, this run's duration; bound by
    VALU issue, and slower than the line's `value` by what the cover saves."""
    t = _load("traffic.json")[w + "_no_cover"]
    k = _kernel(_load(f"{RND}_pmc_{w}_no_cover.json"))
    cyc = k["GRBM_GUI_ACTIVE"]["mean"] / N_XCD
    assert t["valu_issue_frac_guide_clk"] == pytest.approx(k["SQ_INSTS_VALU"]["mean"] * 2.0 / N_SIMD / cyc, abs=6e-4)
    d = _load(f"{RND}_bench_{w}.json")
    r = d["roofline_no_tile_cover"]
    assert r["kernel_ms"] == d["no_tile_cover_ms"] and r["bound"] == "valu_issue"
    assert r["frac"] == t["valu_issue_frac_guide_clk"] and 0.5 < r["frac"] <= 1.0
    assert all(v <= 1.0 for v in r["candidates"].values())
    assert d["no_tile_cover_Mrays_s"] == pytest.approx(PIXELS[w] / (d["no_tile_cover_ms"] * 1e-3) / 1e6)
    assert d["no_tile_cover_ms"] > d["ms_per_step"]


def test_build_line_quotes_the_build_counters():
    t = _load("traffic.json")["c5"]
    d = _load(f"{RND}_bench_c5.json")
    r = d["roofline"]
    assert d["config"]["workload"] == "c5" and d["unit"] == "ms" and d["higher_is_better"] is False
    assert r["traffic"] == pytest.approx(t["build_hbm_bytes"])
    s = d["scene"]
    b_build = 72.0 * d["config"]["triangles"] + 12.0 * 1024 ** 3 + 20.0 * s["refs"]
    assert r["algorithmic_bytes"] == pytest.approx(b_build)
    assert r["frac"] == pytest.approx(b_build / (r["csr_ms"] * 1e-3) / 8e12) and r["frac"] <= 1.0
    # the counters say the CSR's kernels move the algorithmic bytes and little more (no wasted re-reads)
    assert 0.9 < r["traffic"] / b_build < 1.1


def test_config2_line_carries_its_build_counters():
    """Round 2's line had roofline_build.traffic = null for config 2 (no counter pass over its build)."""
    t = _load("traffic.json")["c2"]
    rb = _load(f"{RND}_bench_c2.json")["roofline_build"]
    assert rb["traffic"] == pytest.approx(t["build_hbm_bytes"]) and rb["traffic"] > 0
    assert rb["traffic_source"].startswith("profiles/" + RND + "_pmc_build_c2.json")
    assert 0.0 < rb["frac"] <= 1.0
