"""Copies one collection run (gpurun_out/<name>: bench.json, bench_under_rocprof.json, kt_bench, kt_c3, kt_c5, pmc) into
profiles/ and derives profiles/traffic.json (HBM bytes and busy fractions of k_render) from the counter passes."""
import glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", sys.argv[1])
pm = json.load(open(F + "/pmc/summary.json"))
json.dump(pm, open(R + "/profiles/r1_pmc_c2.json", "w"), indent=1)
k = [v for n, v in pm.items() if "k_render<false, true, false, false>" in n][0]
m = lambda c: k[c]["mean"]
cyc = m("GRBM_GUI_ACTIVE") / 8
fetch, write = m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
t = {"c2": {
    "k_render_hbm_bytes_per_launch": fetch + write, "k_render_hbm_bytes_per_launch_upper": 2 * fetch + write,
    "source": "profiles/r1_pmc_c2.json (FETCH_SIZE + WRITE_SIZE, KB x 1024; upper = 2 x FETCH + WRITE), kernel k_render<false,true,false,false>",
    "valu_issue_frac": round(m("SQ_INSTS_VALU") * 4 / 1024 / cyc, 3), "salu_issue_frac": round(m("SQ_INSTS_SALU") / 256 / cyc, 3),
    "td_busy_frac": round(m("TD_TD_BUSY_sum") / 256 / cyc, 3), "ta_busy_frac": round(m("TA_TA_BUSY_sum") / 256 / cyc, 3),
    "vector_loads_per_launch": m("SQ_INSTS_VMEM_RD"), "valu_insts_per_launch": m("SQ_INSTS_VALU"),
    "salu_insts_per_launch": m("SQ_INSTS_SALU"), "kernel_cycles": cyc,
    "l2_hit_rate": round(m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")), 3),
    "busy_source": "profiles/r1_pmc_c2.json: SQ_INSTS_VALU x 4 clk / 1024 SIMDs, SQ_INSTS_SALU / 256 CUs (one scalar unit per CU), "
                   "TD_TD_BUSY_sum / 256 CUs, over GRBM_GUI_ACTIVE / 8 XCDs"}}
json.dump(t, open(R + "/profiles/traffic.json", "w"), indent=1)
print(json.dumps(t["c2"], indent=0))
for sub, dst in (("kt_bench", "r1_bench_c2_kernel_stats.csv"), ("kt_c3", "r1_render_c3_kernel_stats.csv"), ("kt_c5", "r1_build_c5_kernel_stats.csv")):
    f = glob.glob(F + "/" + sub + "/*/*kernel_stats.csv")
    if f: shutil.copy(f[0], R + "/profiles/" + dst)
for src, dst in (("bench.json", "r1_bench_c2.json"), ("bench_under_rocprof.json", "r1_bench_c2_under_rocprof.json")):
    d = json.loads(open(F + "/" + src).read().strip().splitlines()[-1])
    json.dump(d, open(R + "/profiles/" + dst, "w"), indent=1)
    print(dst, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["voxel_build_ms"])
