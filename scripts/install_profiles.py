"""Copies one collection run (gpurun_out/<name>, written by scripts/collect_profiles.sh) into profiles/ as <round>_* (HVRT_ROUND, default r3) and derives
profiles/traffic.json -- what bench.py's `roofline` objects read.  Two steps, because the bench line quotes traffic.json:
  install_profiles.py <name> pmc      counters -> profiles/<round>_pmc_c2.json, <round>_pmc_build_c5.json, traffic.json
  install_profiles.py <name> bench    bench lines and kernel-trace summaries (collected AFTER the step above)
Every fraction in a bench line can be recomputed from the files this leaves under profiles/ (profiles/README.md)."""
import glob, json, os, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", sys.argv[1])
step = sys.argv[2] if len(sys.argv) > 2 else "pmc"
P = os.path.join(R, "profiles")
RND = os.environ.get("HVRT_ROUND", "r3")
KERNEL = "k_render<false, true, false, false>"
N_SIMD, N_CU, N_XCD = 1024, 256, 8
CLK_WAVE_LOAD_X4 = 16.3        # profiles/r2_ubench_l1_rate.txt: coalesced 16-byte-per-lane wave-load, clk per CU
CLK_VALU_FLOOR = 2.2           # profiles/r2_ubench_valu_rate.txt: a plain VOP1 / VOP2 instruction with >= 2 waves resident
CLK_VALU_GUIDE = 2.0           # MI355X_MICROARCH.md: a wave64 VALU instruction with several waves on the SIMD


def valu_fracs(insts, clk_mix, cyc):
    """VALU issue as a fraction of the SIMDs' clocks.  The floor prices every instruction at the plain rate (no
    instruction issues faster), so it cannot overstate; the mix estimate prices the hot loops' STATIC instruction mix
    (profiles/r2_valu_mix.json) and overstates where the cheap instructions run more often than they are written
    (it exceeds 1 on configs 3 and 4) -- it is reported capped at 1, with the raw figure beside it."""
    raw = insts * clk_mix / N_SIMD / cyc
    return {"valu_issue_frac": round(insts * CLK_VALU_FLOOR / N_SIMD / cyc, 3), "valu_issue_frac_guide_clk": round(insts * CLK_VALU_GUIDE / N_SIMD / cyc, 3),
            "valu_issue_frac_mix": round(min(1.0, raw), 3),
            "valu_issue_frac_mix_raw": round(raw, 3), "valu_clocks_per_inst_floor": CLK_VALU_FLOOR}


def mean_of(k, c):
    return k[c]["mean"]


if step == "pmc":
    pm = json.load(open(F + "/pmc_render/summary.json"))
    json.dump(pm, open(P + "/" + RND + "_pmc_c2.json", "w"), indent=1)
    k = [v for n, v in pm.items() if KERNEL in n][0]
    m = lambda c: mean_of(k, c)
    cyc = m("GRBM_GUI_ACTIVE") / N_XCD
    fetch, write = m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
    mix = json.load(open(P + "/" + RND + "_valu_mix.json"))
    clk_valu = mix["mean_issue_clocks_per_instruction"]
    c2 = {
        "kernel": KERNEL,
        # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE is exact
        "k_render_hbm_bytes_per_launch": 2 * fetch + write,
        "k_render_fetch_size_bytes_raw": fetch, "k_render_write_size_bytes": write,
        "source": "profiles/" + RND + "_pmc_c2.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch of " + KERNEL +
                  "; busy fractions over GRBM_GUI_ACTIVE / 8 XCDs",
        "kernel_cycles": cyc,
        "valu_insts_per_launch": m("SQ_INSTS_VALU"), "valu_clocks_per_inst": clk_valu,
        **valu_fracs(m("SQ_INSTS_VALU"), clk_valu, cyc),
        "salu_insts_per_launch": m("SQ_INSTS_SALU"), "salu_issue_frac": round(m("SQ_INSTS_SALU") / N_CU / cyc, 3),
        "vector_loads_per_launch": m("SQ_INSTS_VMEM_RD"),
        "vector_loads_per_tile": m("SQ_INSTS_VMEM_RD") / (3840 * 2160 / 64),
        "vmem_return_frac": round(m("SQ_INSTS_VMEM_RD") * CLK_WAVE_LOAD_X4 / N_CU / cyc, 3),
        "lds_insts_per_launch": m("SQ_INSTS_LDS") if "SQ_INSTS_LDS" in k else None,
        "td_busy_frac": round(m("TD_TD_BUSY_sum") / N_CU / cyc, 3), "ta_busy_frac": round(m("TA_TA_BUSY_sum") / N_CU / cyc, 3),
        "l2_hit_rate": round(m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")), 3),
        "busy_source": "valu_issue_frac: SQ_INSTS_VALU x 2.2 clk (the plain VOP1 / VOP2 rate, profiles/r2_ubench_valu_rate.txt) / 1024 SIMDs, a floor; "
                       "valu_issue_frac_mix: x the static hot-loop mix's mean issue clocks (profiles/" + RND + "_valu_mix.json), capped at 1; "
                       "SQ_INSTS_VMEM_RD x 16.3 clk (profiles/r2_ubench_l1_rate.txt, coalesced 16 B per lane) / 256 CUs; "
                       "SQ_INSTS_SALU / 256 CUs (one scalar unit per CU)",
    }
    t = {"c2": c2}
    bp = F + "/pmc_build/summary.json"
    if os.path.exists(bp):
        pb = json.load(open(bp))
        json.dump(pb, open(P + "/" + RND + "_pmc_build_c5.json", "w"), indent=1)
        n_builds = 3
        once = ("k_prepare_records", "k_bounds")                  # hvrt_set_mesh / hvrt_mesh_bounds, not the build
        fetch = sum(v["FETCH_SIZE"]["sum"] for n, v in pb.items() if "FETCH_SIZE" in v and not n.startswith(once) and "copyBuffer" not in n)
        write = sum(v["WRITE_SIZE"]["sum"] for n, v in pb.items() if "WRITE_SIZE" in v and not n.startswith(once) and "copyBuffer" not in n)
        atom = sum(v["TCC_ATOMIC_sum"]["sum"] for n, v in pb.items() if "TCC_ATOMIC_sum" in v and not n.startswith(once))
        per = {n: {"fetch_x2_MB": 2 * v["FETCH_SIZE"]["mean"] * 1024 / 1e6, "write_MB": pb[n]["WRITE_SIZE"]["mean"] * 1024 / 1e6,
                   "atomics_M": pb[n].get("TCC_ATOMIC_sum", {"mean": 0})["mean"] / 1e6}
               for n, v in pb.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in pb[n]}
        csr = ("k_count", "k_scan", "k_fill", "k_rank_sort", "k_sort_long", "__amd_rocclr_fillBufferAligned")
        csr_bytes = sum((2 * v["FETCH_SIZE"]["sum"] + pb[n]["WRITE_SIZE"]["sum"]) * 1024 for n, v in pb.items()
                        if n.startswith(csr) and "FETCH_SIZE" in v) / n_builds
        t["c5"] = {"build_hbm_bytes": csr_bytes,        # the CSR's kernels and memsets: what B_build prices
                   "build_with_derived_data_hbm_bytes": (2 * fetch + write) * 1024 / n_builds,
                   "build_fetch_size_bytes_raw": fetch * 1024 / n_builds, "build_write_size_bytes": write * 1024 / n_builds,
                   "build_l2_atomics": atom / n_builds, "build_per_kernel_per_launch": per,
                   "build_source": "profiles/" + RND + "_pmc_build_c5.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the CSR's kernels "
                                   "(k_count, k_scan*, k_fill, k_rank_sort, k_sort_long) and the runtime's fills, per build (3 builds of "
                                   "scripts/prof_build.py c5); build_with_derived_data_hbm_bytes adds distance field, face bits, records"}
    b2 = F + "/pmc_build_c2/summary.json"
    if os.path.exists(b2):
        pb2 = json.load(open(b2))
        json.dump(pb2, open(P + "/" + RND + "_pmc_build_c2.json", "w"), indent=1)
        csr = ("k_count", "k_scan", "k_fill", "k_rank_sort", "k_sort_long", "__amd_rocclr_fillBufferAligned")
        once = ("k_prepare_records", "k_bounds")
        c2["build_hbm_bytes"] = sum((2 * v["FETCH_SIZE"]["sum"] + pb2[n]["WRITE_SIZE"]["sum"]) * 1024 for n, v in pb2.items()
                                    if n.startswith(csr) and "FETCH_SIZE" in v) / 3
        c2["build_with_derived_data_hbm_bytes"] = sum((2 * v["FETCH_SIZE"]["sum"] + pb2[n]["WRITE_SIZE"]["sum"]) * 1024 for n, v in pb2.items()
                                                      if "FETCH_SIZE" in v and not n.startswith(once) and "copyBuffer" not in n) / 3
        c2["build_source"] = "profiles/" + RND + "_pmc_build_c2.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the CSR's kernels and the runtime's fills, per build (3 builds of scripts/prof_build.py c2)"
    for w, px in (("c3", 3840 * 2160), ("c4", 7680 * 4320)):
        fp = F + "/pmc_render_" + w + "/summary.json"
        if not os.path.exists(fp):
            continue
        pw = json.load(open(fp))
        json.dump(pw, open(P + "/" + RND + "_pmc_" + w + ".json", "w"), indent=1)
        kw = [v for n, v in pw.items() if KERNEL in n][0]
        mw = lambda c: mean_of(kw, c)
        cw = mw("GRBM_GUI_ACTIVE") / N_XCD
        t[w] = {"kernel": KERNEL,
                "k_render_hbm_bytes_per_launch": (2 * mw("FETCH_SIZE") + mw("WRITE_SIZE")) * 1024,
                "kernel_cycles": cw, "valu_insts_per_launch": mw("SQ_INSTS_VALU"), "valu_clocks_per_inst": clk_valu,
                **valu_fracs(mw("SQ_INSTS_VALU"), clk_valu, cw),
                "salu_issue_frac": round(mw("SQ_INSTS_SALU") / N_CU / cw, 3),
                "vector_loads_per_launch": mw("SQ_INSTS_VMEM_RD"), "vector_loads_per_tile": mw("SQ_INSTS_VMEM_RD") / (px / 64),
                "vmem_return_frac": round(mw("SQ_INSTS_VMEM_RD") * CLK_WAVE_LOAD_X4 / N_CU / cw, 3),
                "source": "profiles/" + RND + "_pmc_" + w + ".json, priced like c2 (the hot loops are the same code: profiles/" + RND + "_valu_mix.json)"}
        print(w, {k: v for k, v in t[w].items() if not isinstance(v, str)})
    # the same kernel with every tile walked (HVRT_TILE_COVER=0): where it is bound by throughput, not by its longest tiles
    for w, sub, px in (("c2", "pmc_render_no_cover", 3840 * 2160), ("c3", "pmc_render_c3_no_cover", 3840 * 2160), ("c4", "pmc_render_c4_no_cover", 7680 * 4320)):
        fp = F + "/" + sub + "/summary.json"
        if not os.path.exists(fp):
            continue
        pw = json.load(open(fp))
        json.dump(pw, open(P + "/" + RND + "_pmc_" + w + "_no_cover.json", "w"), indent=1)
        kw = [v for n, v in pw.items() if KERNEL in n][0]
        mw = lambda c: mean_of(kw, c)
        cw = mw("GRBM_GUI_ACTIVE") / N_XCD
        t[w + "_no_cover"] = {"kernel": KERNEL + ", HVRT_TILE_COVER=0",
                "k_render_hbm_bytes_per_launch": (2 * mw("FETCH_SIZE") + mw("WRITE_SIZE")) * 1024,
                "kernel_cycles": cw, "valu_insts_per_launch": mw("SQ_INSTS_VALU"), "valu_clocks_per_inst": clk_valu,
                **valu_fracs(mw("SQ_INSTS_VALU"), clk_valu, cw),
                "salu_issue_frac": round(mw("SQ_INSTS_SALU") / N_CU / cw, 3),
                "vector_loads_per_launch": mw("SQ_INSTS_VMEM_RD"), "vector_loads_per_tile": mw("SQ_INSTS_VMEM_RD") / (px / 64),
                "vmem_return_frac": round(mw("SQ_INSTS_VMEM_RD") * CLK_WAVE_LOAD_X4 / N_CU / cw, 3),
                "source": "profiles/" + RND + "_pmc_" + w + "_no_cover.json: the counter passes with HVRT_TILE_COVER=0 (every tile walked), priced like c2"}
        if "TCC_HIT_sum" in kw:
            t[w + "_no_cover"]["l2_hit_rate"] = round(mw("TCC_HIT_sum") / (mw("TCC_HIT_sum") + mw("TCC_MISS_sum")), 3)
        print(w + "_no_cover", {k: v for k, v in t[w + "_no_cover"].items() if not isinstance(v, str)})
    json.dump(t, open(P + "/traffic.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in c2.items() if not isinstance(v, str)}, indent=0))
    if "c5" in t:
        print({k: v for k, v in t["c5"].items() if not isinstance(v, (str, dict))})
else:
    for sub, dst in (("kt_bench", RND + "_bench_c2_kernel_stats.csv"), ("kt_c3", RND + "_bench_c3_kernel_stats.csv"),
                     ("kt_c4", RND + "_bench_c4_kernel_stats.csv"), ("kt_c5", RND + "_build_c5_kernel_stats.csv")):
        f = sorted(glob.glob(F + "/" + sub + "/*/*kernel_stats.csv"), key=os.path.getmtime)     # a re-run leaves older files behind
        if f:
            shutil.copy(f[-1], P + "/" + dst)
    for src, dst in (("bench_c2.json", RND + "_bench_c2.json"), ("bench_c2_under_rocprof.json", RND + "_bench_c2_under_rocprof.json"),
                     ("bench_c3.json", RND + "_bench_c3.json"), ("bench_c4.json", RND + "_bench_c4.json"), ("bench_c5.json", RND + "_bench_c5.json"),
                     ("bench_c2_moving_camera.json", RND + "_bench_c2_moving_camera.json")):
        if not os.path.exists(F + "/" + src):
            continue
        d = json.loads(open(F + "/" + src).read().strip().splitlines()[-1])
        json.dump(d, open(P + "/" + dst, "w"), indent=1)
        rf = d.get("roofline", {})
        print(dst, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "roofline", rf.get("bound"), rf.get("frac"),
              "build", d.get("voxel_build_ms"))
