"""Prices the vector-ALU instructions of a kernel's hot loops with the measured issue costs of scripts/ubench/valu_rate.hip.

  python scripts/valu_mix.py [kernel-name-substring]  > profiles/rN_valu_mix.json

The counters give SQ_INSTS_VALU (how many vector-ALU wave-instructions a launch executes) but nothing that says
how long a SIMD is busy issuing them (SQ_ACTIVE_INST_VALU equals the instruction count whatever the instruction:
checked against the micro-benchmark, profiles/r2_ubench_valu_rate.txt).  On gfx950 the cost depends on the encoding:
2 clocks for a plain VOP1/VOP2 instruction, 4 for VOP3 / VOPC / DPP / readlane, 8 for the transcendental ones
(measured with two or more waves on the SIMD).  This script compiles hvrt_trace.hip to assembly (no GPU needed),
takes the instructions of the named kernel that sit in loops of depth >= 3 (the walk, the list loop, the tests --
where the instructions of a launch are executed) and reports their mix and mean cost, which bench.py multiplies by
SQ_INSTS_VALU for the VALU-issue side of its roofline.  A static mix, not a dynamic one: it weights every hot-loop
instruction equally."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "raytracer_amd", "csrc")
kernel = sys.argv[1] if len(sys.argv) > 1 else "k_renderILb0ELb1ELb0ELb0E"
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt " \
        "-fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -S --cuda-device-only"
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "t.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS.split(), "-I" + os.path.join(ROOT, "include"), os.path.join(CS, "hvrt_trace.hip"), "-o", out],
                   check=True, capture_output=True)
    lines = open(out).read().splitlines()

TRANS = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
def cost(op, text):
    if op.startswith(TRANS): return 8, "transcendental (8)"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return 4, "lane access (4)"
    if op.startswith("v_cmp"): return 4, "compare (4)"
    if "dpp" in text or "quad_perm" in text or "row_" in text: return 4, "DPP (4)"
    if op.endswith("_e64") or op.startswith(("v_fma", "v_mad", "v_add3", "v_or3", "v_and_or", "v_lshl_add", "v_lshl_or", "v_alignbit", "v_bfe", "v_bfi",
                                            "v_div_", "v_mul_lo", "v_mul_hi", "v_cvt", "v_xad", "v_add_lshl", "v_perm", "v_min3", "v_max3", "v_med3",
                                            "v_pk_", "v_mbcnt")): return 4, "VOP3 (4)"
    if op.startswith("v_cndmask"): return 2, "select on a VALU-written vcc (2)"
    return 2, "VOP1 / VOP2 (2)"

inside, depth, mix, n, clk = False, 0, {}, 0, 0
in_asm = False
for l in lines:
    if re.match(r"^_Z\w+:", l):
        inside = kernel in l
        depth = 0
        continue
    if not inside:
        continue
    s = l.strip()
    if s.startswith("s_endpgm"):
        inside = False
        continue
    m = re.search(r"Depth=(\d+)", s)
    if m and ("Loop Header" in s or "in Loop" in s):
        depth = int(m.group(1))
    if not s.startswith("v_"):
        continue
    op = s.split()[0]
    if depth >= 3:
        c, k = cost(op, s)
        mix[k] = mix.get(k, 0) + 1
        n += 1
        clk += c
print(json.dumps({"kernel": kernel, "what": "VALU instructions in loops of depth >= 3, static count, priced with profiles/r2_ubench_valu_rate.txt",
                  "instructions": n, "mix": mix, "mean_issue_clocks_per_instruction": round(clk / max(n, 1), 3)}, indent=1))
