"""Copies one collection run (gpurun_out/<name>, written by scripts/collect_profiles.sh) into profiles/ as r2_* and derives
profiles/traffic.json -- what bench.py's `roofline` objects read.  Two steps, because the bench line quotes traffic.json:
  install_profiles.py <name> pmc      counters -> profiles/r2_pmc_c2.json, r2_pmc_build_c5.json, traffic.json
  install_profiles.py <name> bench    bench lines and kernel-trace summaries (collected AFTER the step above)
Every fraction in a bench line can be recomputed from the files this leaves under profiles/ (profiles/README.md)."""
import glob, json, os, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", sys.argv[1])
step = sys.argv[2] if len(sys.argv) > 2 else "pmc"
P = os.path.join(R, "profiles")
KERNEL = "k_render<false, true, false, false>"
N_SIMD, N_CU, N_XCD = 1024, 256, 8
CLK_WAVE_LOAD_X4 = 16.3        # profiles/r2_ubench_l1_rate.txt: coalesced 16-byte-per-lane wave-load, clk per CU


def mean_of(k, c):
    return k[c]["mean"]


if step == "pmc":
    pm = json.load(open(F + "/pmc_render/summary.json"))
    json.dump(pm, open(P + "/r2_pmc_c2.json", "w"), indent=1)
    k = [v for n, v in pm.items() if KERNEL in n][0]
    m = lambda c: mean_of(k, c)
    cyc = m("GRBM_GUI_ACTIVE") / N_XCD
    fetch, write = m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
    mix = json.load(open(P + "/r2_valu_mix.json"))
    clk_valu = mix["mean_issue_clocks_per_instruction"]
    c2 = {
        "kernel": KERNEL,
        # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE is exact
        "k_render_hbm_bytes_per_launch": 2 * fetch + write,
        "k_render_fetch_size_bytes_raw": fetch, "k_render_write_size_bytes": write,
        "source": "profiles/r2_pmc_c2.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch of " + KERNEL +
                  "; busy fractions over GRBM_GUI_ACTIVE / 8 XCDs",
        "kernel_cycles": cyc,
        "profiled_kernel_ms": cyc / 2.4e6,
        "valu_insts_per_launch": m("SQ_INSTS_VALU"), "valu_clocks_per_inst": clk_valu,
        "valu_issue_frac": round(m("SQ_INSTS_VALU") * clk_valu / N_SIMD / cyc, 3),
        "valu_issue_frac_at_2clk": round(m("SQ_INSTS_VALU") * 2.0 / N_SIMD / cyc, 3),
        "valu_issue_frac_at_4clk": round(m("SQ_INSTS_VALU") * 4.0 / N_SIMD / cyc, 3),
        "salu_insts_per_launch": m("SQ_INSTS_SALU"), "salu_issue_frac": round(m("SQ_INSTS_SALU") / N_CU / cyc, 3),
        "vector_loads_per_launch": m("SQ_INSTS_VMEM_RD"),
        "vector_loads_per_tile": m("SQ_INSTS_VMEM_RD") / (3840 * 2160 / 64),
        "vmem_return_frac": round(m("SQ_INSTS_VMEM_RD") * CLK_WAVE_LOAD_X4 / N_CU / cyc, 3),
        "lds_insts_per_launch": m("SQ_INSTS_LDS") if "SQ_INSTS_LDS" in k else None,
        "td_busy_frac": round(m("TD_TD_BUSY_sum") / N_CU / cyc, 3), "ta_busy_frac": round(m("TA_TA_BUSY_sum") / N_CU / cyc, 3),
        "l2_hit_rate": round(m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")), 3),
        "busy_source": "SQ_INSTS_VALU x mean issue clocks (profiles/r2_valu_mix.json, priced by profiles/r2_ubench_valu_rate.txt) / 1024 SIMDs; "
                       "SQ_INSTS_VMEM_RD x 16.3 clk (profiles/r2_ubench_l1_rate.txt, coalesced 16 B per lane) / 256 CUs; "
                       "SQ_INSTS_SALU / 256 CUs (one scalar unit per CU)",
    }
    t = {"c2": c2}
    bp = F + "/pmc_build/summary.json"
    if os.path.exists(bp):
        pb = json.load(open(bp))
        json.dump(pb, open(P + "/r2_pmc_build_c5.json", "w"), indent=1)
        n_builds = 3
        once = ("k_prepare_records", "k_bounds")                  # hvrt_set_mesh / hvrt_mesh_bounds, not the build
        fetch = sum(v["FETCH_SIZE"]["sum"] for n, v in pb.items() if "FETCH_SIZE" in v and not n.startswith(once) and "copyBuffer" not in n)
        write = sum(v["WRITE_SIZE"]["sum"] for n, v in pb.items() if "WRITE_SIZE" in v and not n.startswith(once) and "copyBuffer" not in n)
        atom = sum(v["TCC_ATOMIC_sum"]["sum"] for n, v in pb.items() if "TCC_ATOMIC_sum" in v and not n.startswith(once))
        per = {n: {"fetch_x2_MB": 2 * v["FETCH_SIZE"]["mean"] * 1024 / 1e6, "write_MB": pb[n]["WRITE_SIZE"]["mean"] * 1024 / 1e6,
                   "atomics_M": pb[n].get("TCC_ATOMIC_sum", {"mean": 0})["mean"] / 1e6}
               for n, v in pb.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in pb[n]}
        t["c5"] = {"build_hbm_bytes": (2 * fetch + write) * 1024 / n_builds,
                   "build_fetch_size_bytes_raw": fetch * 1024 / n_builds, "build_write_size_bytes": write * 1024 / n_builds,
                   "build_l2_atomics": atom / n_builds, "build_per_kernel_per_launch": per,
                   "build_source": "profiles/r2_pmc_build_c5.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every kernel and "
                                   "runtime fill of a build (3 builds of scripts/prof_build.py c5; set_mesh's kernels excluded)"}
    json.dump(t, open(P + "/traffic.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in c2.items() if not isinstance(v, str)}, indent=0))
    if "c5" in t:
        print({k: v for k, v in t["c5"].items() if not isinstance(v, (str, dict))})
else:
    for sub, dst in (("kt_bench", "r2_bench_c2_kernel_stats.csv"), ("kt_c3", "r2_bench_c3_kernel_stats.csv"),
                     ("kt_c4", "r2_bench_c4_kernel_stats.csv"), ("kt_c5", "r2_build_c5_kernel_stats.csv")):
        f = glob.glob(F + "/" + sub + "/*/*kernel_stats.csv")
        if f:
            shutil.copy(f[0], P + "/" + dst)
    for src, dst in (("bench_c2.json", "r2_bench_c2.json"), ("bench_c2_under_rocprof.json", "r2_bench_c2_under_rocprof.json"),
                     ("bench_c3.json", "r2_bench_c3.json"), ("bench_c4.json", "r2_bench_c4.json"), ("bench_c5.json", "r2_bench_c5.json")):
        if not os.path.exists(F + "/" + src):
            continue
        d = json.loads(open(F + "/" + src).read().strip().splitlines()[-1])
        json.dump(d, open(P + "/" + dst, "w"), indent=1)
        rf = d.get("roofline", {})
        print(dst, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "roofline", rf.get("bound"), rf.get("frac"),
              "build", d.get("voxel_build_ms"))
