"""Fold rocprofv3 `--pmc ... --output-format csv` outputs (one *_counter_collection.csv per pass) into per-kernel
means: {kernel short name: {counter: mean value per dispatch, "dispatches": n, "mean_ns": t}}.
usage: pmc_summary.py OUT.json FILTER_SUBSTRING[,FILTER2...] DIR_OR_CSV [DIR_OR_CSV ...]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    out, filt, srcs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    files = []
    for s in srcs:
        files += [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if not any(x in k for x in filt):
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    res = {}
    for k, cs in acc.items():
        res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        res[k]["dispatches_sampled"] = max(len(v) for v in cs.values())
        res[k]["mean_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
