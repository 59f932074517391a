"""Vector-memory-path counters of the field kernels: TA / TCP / TCC traffic next to the SQ view.

    python tools/pmc_path.py OUT_PREFIX [--config trevi] [extra bench args]      (GPU box, repo root)

`rocprofv3 -L` is asked which of a wish-list of counters this build exposes; the ones that exist are collected in passes of
at most four per hardware block (separate runs: no trace domain beside --pmc), over a short bench run, and averaged per
dispatch of every `field16_*` kernel.  Output: OUT_PREFIX.json (raw averages) and OUT_PREFIX.md (derived table):
    * wave-level: vector-memory read / write instructions per launch (each moves <= 1 KiB per wave);
    * TA busy share, TCP -> TCC read / write requests, TCP stall cycles, TCC requests / hits / misses;
    * bytes per clock and CU = requests x line size / (GRBM_GUI_ACTIVE / 8 XCDs) / 256 CUs.
This process never touches the GPU (it only starts rocprofv3 children), so the profiler's preload is exec-safe."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

WISH = {
    "SQ": ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_VMEM",
           "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL",
           "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY"],
    "TA": ["TA_TA_BUSY_sum", "TA_BUSY_avr", "TA_BUSY_max", "TA_FLAT_READ_WAVEFRONTS_sum", "TA_FLAT_WRITE_WAVEFRONTS_sum",
           "TA_BUFFER_WAVEFRONTS_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum",
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum", "TA_FLAT_WAVEFRONTS_sum"],
    "TD": ["TD_TD_BUSY_sum", "TD_TC_STALL_sum", "TD_LOAD_WAVEFRONT_sum", "TD_STORE_WAVEFRONT_sum"],
    "TCP": ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum",
            "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum",
            "TCP_TD_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum",
            "TCP_TOTAL_ACCESSES_sum", "TCP_TOTAL_READ_sum", "TCP_TOTAL_WRITE_sum", "TCP_TCC_NC_READ_REQ_sum",
            "TCP_TCC_UC_READ_REQ_sum", "TCP_TCC_CC_READ_REQ_sum", "TCP_TCC_RW_READ_REQ_sum"],
    "TCC": ["TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum",
            "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_TAG_STALL_sum", "TCC_BUSY_sum", "TCC_NORMAL_WRITEBACK_sum",
            "TCC_STREAMING_REQ_sum"],
}
PER_PASS = {"SQ": 7, "TA": 2, "TD": 2, "TCP": 3, "TCC": 3}


def main():
    out = os.path.abspath(sys.argv[1])
    extra = sys.argv[2:]
    root = os.getcwd()
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp", env=env)
    text = avail.stdout + avail.stderr
    names = set(re.findall(r"\b((?:SQ|TA|TD|TCP|TCC|GRBM|TCA)_[A-Za-z0-9_]+)\b", text))
    open(out + ".avail.txt", "w").write("\n".join(sorted(names)) + "\n")
    passes = []
    for blk, wish in WISH.items():
        have = [c for c in wish if c in names]
        n = PER_PASS[blk]
        for i in range(0, len(have), n):
            passes.append(have[i:i + n])
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    log = []
    for i, p in enumerate(passes):
        d = f"{out}.pass{i}"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", "GRBM_GUI_ACTIVE", *p, "--output-format", "csv", "-d", d, "-o", "p", "--",
               "python3", os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing",
               "--no-extras", "--no-configs34", *extra]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env, timeout=900)
        log.append(f"pass {i} {p} rc={r.returncode}\n{r.stderr[-600:]}")
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "field16" not in k:
                    continue
                k = re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", k))
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k][row["Counter_Name"]].add((i, row["Dispatch_Id"]))  # GRBM_GUI_ACTIVE rides in every pass
        shutil.rmtree(d, ignore_errors=True)
    open(out + ".log", "w").write("\n".join(log))
    res = {k: {c: acc[k][c] / len(disp[k][c]) for c in acc[k]} for k in acc}
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        for k, v in res.items():
            cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8  # summed over the 8 XCDs
            f.write(f"### `{k}`  ({cyc:.4g} cycles per launch)\n\n| quantity | per launch | per clock and CU |\n|---|---|---|\n")

            def row(name, val, per=None):
                f.write(f"| {name} | {val:.4g} | {'' if per is None else f'{per:.3g}'} |\n")
            cu = cyc * 256 if cyc else float("nan")
            for c, label, b in (("SQ_INSTS_VMEM_RD", "vector-memory read instructions (<= 1 KiB each)", 1024),
                                ("SQ_INSTS_VMEM_WR", "vector-memory write instructions (<= 1 KiB each)", 1024),
                                ("TCP_TCC_READ_REQ_sum", "TCP -> TCC read requests (x 64 B)", 64),
                                ("TCP_TCC_WRITE_REQ_sum", "TCP -> TCC write requests (x 64 B)", 64),
                                ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP cache-line accesses (x 128 B)", 128),
                                ("TCC_REQ_sum", "TCC requests (x 128 B)", 128), ("TCC_EA0_RDREQ_sum", "TCC -> fabric read requests (x 64 B)", 64),
                                ("TCC_EA0_WRREQ_sum", "TCC -> fabric write requests (x 64 B)", 64)):
                if c in v:
                    row(label + f" [{c}]", v[c], v[c] * b / cu)
            for c in sorted(v):
                if any(s in c for s in ("BUSY", "STALL", "FIFO_FULL", "GATE_EN", "CYCLES_VMEM", "ACTIVE_INST_VMEM")) and c != "GRBM_GUI_ACTIVE":
                    # *_sum counters add up 256 CU instances (TA / TCP / TD) or 128 TCC channels (16 per XCD); SQ ones per-wave cycles
                    inst = 128 if c.startswith("TCC_") else 256
                    row(f"{c} (share of {inst} x cycles)", v[c], v[c] / (cyc * inst) if cyc else float("nan"))
            if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
                row("L2 hit rate", v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]))
            f.write("\n")
    print(open(out + ".md").read())
    print(open(out + ".log").read()[-3000:])


if __name__ == "__main__":
    main()
