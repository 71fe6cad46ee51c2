"""Developer tool: merge rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<name>.json.
usage: python tools/pmc_summary.py out.json workload=<fetch_dir>,<write_dir> [workload=...]
Per kernel (template arguments kept, parameter list dropped): average FETCH_SIZE / WRITE_SIZE per dispatch in KB as
reported, and the HBM bytes per dispatch with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of
the bytes of 16-B-per-lane streaming reads: doubled)."""
import csv, glob, json, os, sys, collections


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0]
                a = acc[k]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averages per dispatch in KB as reported; gfx950 "
                "correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane streaming reads -> "
                "doubled in hbm_bytes_corrected",
       "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python bench.py --no-cpu-baseline [--workload cube64]"}
for spec in sys.argv[2:]:
    name, dirs = spec.split("=")
    fd, wd = dirs.split(",")
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    ent = {}
    for k in sorted(F, key=lambda k: -F[k][0] * F[k][1]):
        if not k.startswith(("void k_", "k_")):
            continue
        f, n = F[k]
        w = W.get(k, (0.0, 0))[0]
        ent[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": (2.0 * f + w) * 1024.0, "dispatches": n}
    out[name] = ent
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], {k: len(v) for k, v in out.items() if isinstance(v, dict)})
