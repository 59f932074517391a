"""Write profiles/pmc_current.json: the HBM bytes per launch bench.py puts into `roofline.traffic`, tagged with the hash
of the HIP sources they were measured on (bench.py ignores the file when the hash, field mode, config or progress differ).

    python tools/pmc_current.py profiles/r02_pmc.json --field f16x3 --config brandenburg --progress 0.3
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("pmc_json")
ap.add_argument("--field", default="f16x3")
ap.add_argument("--config", default="brandenburg")
ap.add_argument("--progress", type=float, default=0.3)
ap.add_argument("--out", default="pmc_current.json", help="file name under profiles/ (bench.py reads pmc_current.json for the headline, pmc_current_trevi.json for the trevi object)")
ap.add_argument("--name", default=None, help="name the PMC json is committed under in profiles/ (e.g. r06_pmc.json): what `roofline.traffic_source` cites; default: the input file's name")
a = ap.parse_args()
pmc = json.load(open(a.pmc_json))
np_ = "1" if a.field == "f16" else "2"
rr = a.field == "f16" and "rr16_fwd_kernel<8>" in pmc  # the fp16 mode's register-resident kernels (csrc/field16rr.hip)
pick = {"field_fwd": "rr16_fwd_kernel<8>" if rr else (f"field16_fwd_kernel<{np_}, 64, 4>" if a.field != "f32" else "field_fwd_kernel<256, 64>"),
        "field_bwd": "rr16_bwd_kernel<8>" if rr else (f"field16_bwd_kernel<{np_}, 64, 4>" if a.field != "f32" else "field_bwd_kernel<256, 64>"),
        "wgrad16_256x256": f"wgrad_f16x3_kernel<{np_}, 4, 4>", "wgrad16p_256x256": "wgrad_f16p_kernel<4, 4, 1, 1>" if rr else "wgrad_f16p_kernel<4, 4, 1, 0>",
        "wgrad_256x256": "wgrad_kernel<4, 4>"}
kern = {}
for name, key in pick.items():
    t = pmc.get(key)
    if t and "fetch_bytes_per_launch" in t and "write_bytes_per_launch" in t:
        kern[name] = {"kernel": key, "fetch_bytes_per_launch": t["fetch_bytes_per_launch"],
                      "write_bytes_per_launch": t["write_bytes_per_launch"], "mfma_busy": t.get("mfma_busy")}
# all kernels of one step: (FETCH x2 + WRITE) x dispatches, over the steps the PMC run made (two field forward launches each)
fwd = pmc.get(pick["field_fwd"], {})
steps = fwd.get("dispatches", 0) / 2
step_bytes = None
if steps:
    step_bytes = sum((t.get("fetch_bytes_per_launch", 0) + t.get("write_bytes_per_launch", 0)) * t.get("dispatches", 0)
                     for t in pmc.values() if isinstance(t, dict)) / steps
out = {"src_sha16": bench.source_sha16(), "hbm_bytes_per_step": step_bytes, "field": a.field, "config": a.config, "progress": a.progress,
       "file": a.name or os.path.basename(a.pmc_json), "kernels": kern}
json.dump(out, open(os.path.join(ROOT, "profiles", a.out), "w"), indent=1)
print(json.dumps(out, indent=1))
