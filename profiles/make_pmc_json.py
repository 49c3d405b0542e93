"""profiles/pmc_knn.json (what bench.py reports as roofline.traffic) from the three PMC passes of scripts/profile_round6.sh: usage make_pmc_json.py <dir with pmc1..3.txt> <tag>.
HBM bytes per launch = 2 x FETCH_SIZE (KB; gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE (KB), means over the dispatches."""
import json, os, re, sys
d, tag = sys.argv[1], sys.argv[2]
rows = {}
for i in (1, 2, 3):
    for line in open(os.path.join(d, f"pmc{i}.txt")):
        m = re.match(r"^(.*?)\s+([A-Z][A-Za-z_0-9]+)\s+(\d+)\s+([0-9.eE+-]+)\s*$", line.rstrip())
        if m and m.group(2) != "dispatches":
            rows[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))

def get(kern_sub, counter):
    for (k, c), v in rows.items():
        if c == counter and kern_sub in k:
            return v
    return (0, None)

def hbm(kern_sub):
    f, w = get(kern_sub, "FETCH_SIZE")[1], get(kern_sub, "WRITE_SIZE")[1]
    return None if f is None or w is None else int(round((2.0 * f + w) * 1024.0))
main = "knn_features_kernel<0, false, false, 1, true"       # (prefix: round 5 added a template parameter behind these)
cold = "knn_features_kernel<0, false, false, 0, false"
fit = "fit_linearize_kernel<5, false, false"
sq = {c: get(main, "SQ_" + c)[1] for c in ("WAVES", "BUSY_CYCLES", "INSTS_VALU", "INSTS_SALU", "INSTS_LDS", "WAVE_CYCLES", "WAIT_ANY", "WAIT_INST_ANY")}
valu = None
if sq["INSTS_VALU"] and sq["BUSY_CYCLES"]:
    valu = round(sq["INSTS_VALU"] * 4.0 / 32.0 / sq["BUSY_CYCLES"], 3)
out = {
    "workload": f"2x64_vs_500k_{tag}",
    "kernel": f"mlh::{main} (iterations >= 1 of a solve: the previous iteration's finish in every workgroup, then the bounded search)",
    "source": f"rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum / --pmc SQ_* (separate passes, --kernel-trace only; scripts/profile_round6.sh), "
              f"python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-supplementary --profile-events 0 --synchronous; means over {get(main, 'FETCH_SIZE')[0]} dispatches; FETCH_SIZE doubled per "
              f"MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); see profiles/{tag}_pmc_summary.txt",
    "fetch_size_kb_raw": get(main, "FETCH_SIZE")[1], "write_size_kb": get(main, "WRITE_SIZE")[1],
    "tcc_hit": get(main, "TCC_HIT_sum")[1], "tcc_miss": get(main, "TCC_MISS_sum")[1],
    "hbm_bytes_per_launch": hbm(main),
    "search_only_kernel": {"kernel": f"mlh::{cold} (iteration 0 of a synchronous solve: the cold search alone)", "fetch_size_kb_raw": get(cold, "FETCH_SIZE")[1],
                           "write_size_kb": get(cold, "WRITE_SIZE")[1], "tcc_hit": get(cold, "TCC_HIT_sum")[1], "tcc_miss": get(cold, "TCC_MISS_sum")[1], "hbm_bytes_per_launch": hbm(cold)},
    "fit_kernel": {"kernel": f"mlh::{fit}", "fetch_size_kb_raw": get(fit, "FETCH_SIZE")[1], "write_size_kb": get(fit, "WRITE_SIZE")[1], "hbm_bytes_per_launch": hbm(fit)},
    "sq_per_shader_engine": {k.lower(): v for k, v in sq.items()},
    "valu_issue_utilisation": valu,
    "valu_insts_per_launch": (int(sq["INSTS_VALU"] * 32) if sq["INSTS_VALU"] else None),
}
print(json.dumps(out, indent=1))
