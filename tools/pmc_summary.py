"""Summarise rocprofv3 --pmc passes (CSV) per kernel.

    python tools/pmc_summary.py OUT.json OUT.md DIR_A [DIR_B ...]

Each DIR holds one pass (`rocprofv3 --pmc <counters> --output-format csv -d DIR ...`).  Values are averaged per dispatch.
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests as
64 bytes, /opt/skills/guides/MI355X_MICROARCH.md section HBM), WRITE_SIZE is exact."""
import collections, csv, glob, json, re, sys

out_json, out_md, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k, c = r["Kernel_Name"], r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"])
            disp[k][c].add(r["Dispatch_Id"])


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    return re.sub(r"\(.*", "", k)


res = {}
for k in acc:
    n = {c: len(disp[k][c]) for c in acc[k]}
    v = {c: acc[k][c] / n[c] for c in acc[k]}
    o = {"dispatches": max(n.values())}
    if "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        o["gpu_cycles"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            o["mfma_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc  # cycles summed over 1024 SIMDs
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            for c, name in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"),
                            ("SQ_ACTIVE_INST_ANY", "active_inst_any"), ("SQ_ACTIVE_INST_VALU", "active_valu"),
                            ("SQ_ACTIVE_INST_LDS", "active_lds"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict"),
                            ("SQ_ACTIVE_INST_VMEM", "active_vmem"), ("SQ_INST_CYCLES_VMEM", "inst_cycles_vmem")):
                if c in v:
                    o[name + "_frac_of_wave_cycles"] = v[c] / wc
    if "FETCH_SIZE" in v:
        o["fetch_bytes_per_launch"] = v["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in v:
        o["write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
    o["raw"] = v
    res[short(k)] = o
keys = sorted(res, key=lambda k: -res[k].get("gpu_cycles", 0) * res[k]["dispatches"])
json.dump({k: res[k] for k in keys}, open(out_json, "w"), indent=1)
with open(out_md, "w") as f:
    cols = ["dispatches", "gpu_cycles", "mfma_busy", "wait_any_frac_of_wave_cycles", "wait_inst_any_frac_of_wave_cycles",
            "active_valu_frac_of_wave_cycles", "active_lds_frac_of_wave_cycles", "lds_bank_conflict_frac_of_wave_cycles",
            "fetch_bytes_per_launch", "write_bytes_per_launch"]
    f.write("| kernel | " + " | ".join(c.replace("_frac_of_wave_cycles", "/wave_cyc") for c in cols) + " |\n")
    f.write("|---|" + "---|" * len(cols) + "\n")
    for k in keys[:14]:
        row = []
        for c in cols:
            x = res[k].get(c)
            row.append("" if x is None else (f"{x:.3g}" if isinstance(x, float) else str(x)))
        f.write(f"| `{k[:48]}` | " + " | ".join(row) + " |\n")
print(open(out_md).read())
