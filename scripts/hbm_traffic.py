"""Combine the two PMC passes (scripts/prof_pmc.sh ... FETCH_SIZE / WRITE_SIZE) into profiles/<round>_unet_hbm_traffic.json.

usage: python scripts/hbm_traffic.py <fetch.csv> <write.csv> <out.json> "<command that was profiled>"
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B and is doubled (MI355X_MICROARCH.md, HBM)."""
import csv, json, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[short(r["Kernel_Name"])] = (int(r["Dispatches"]), float(r["AveragePerDispatch"]))
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
kernels = {}
for k in sorted(set(fetch) & set(write)):
    if not (k.startswith("conv") or k.startswith("tile_")):
        continue
    n, f = fetch[k]; _, w = write[k]
    kernels[k] = {"dispatches": n, "fetch_KiB_raw": f, "write_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
json.dump({"_comment": __doc__.split("\n\n")[1].strip(), "command": sys.argv[4], "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(kernels, indent=1)[:2000])
