"""Scans gfx950 assembly (hipcc -S --cuda-device-only) for VOP2 v_cndmask_b32 instructions that read a vcc last
written by the scalar unit: on MI355X each of them blocks its SIMD for ~19 clocks (scripts/ubench/valu_rate.hip:
's_mov vcc + v_cndmask vcc'), against 2 after a v_cmp and 4 in the VOP3 (_e64) encoding.
  python scripts/vcc_hazard_scan.py file.s [kernel-name-substring]"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
kern, last, depth = None, None, 0
stats = {}
for i, l in enumerate(txt):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern, last = m.group(1), None
        continue
    if kern is None or filt not in kern:
        continue
    s = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", s):
        last = ("label", i)          # unknown across a join
        continue
    if "Loop Header" in s or "Inner Loop" in s:
        dm = re.search(r"Depth=(\d+)", s)
        if dm: depth = int(dm.group(1))
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    if op.startswith("v_cndmask_b32_e32") or (op == "v_cndmask_b32" and s.rstrip().endswith("vcc")):
        kind = "unknown(join)" if last is None or last[0] == "label" else last[0]
        stats.setdefault(kern, {}).setdefault(kind, []).append((i + 1, depth))
    # who writes vcc?
    ops = s.split(None, 1)[1] if " " in s else ""
    dst = ops.split(",")[0].strip() if ops else ""
    if dst in ("vcc", "vcc_lo", "vcc_hi") or (op.startswith("v_cmp") and op.endswith("_e32")):
        last = ("salu" if op.startswith("s_") else "valu", i)
    elif re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", op) and ", vcc," in s:
        last = ("valu", i)
for k, d in stats.items():
    print(k[:90])
    for kind, v in d.items():
        print(f"   vcc last written by {kind}: {len(v)}   lines/depths: {[(a, b) for a, b in v][:40]}")
