"""Fold the counter passes of tools/pmc_traffic.sh into profiles/r03/pmc_traffic.json: HBM bytes per launch for the kernels the
bench line prices, with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) counts HALF of a
wide (>= 128 B per request) coalesced read and 64-byte requests 1:1; WRITE_SIZE (KB) is exact.  Which reads are wide is a
property of the kernel's access pattern, stated per kernel below.
usage: pmc_fold.py PMC_DIR OUT.json"""
import csv, glob, json, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name[:60]


def load(d):
    """-> {kernel: {counter: [values per dispatch, in dispatch order]}}"""
    acc = defaultdict(lambda: defaultdict(list))
    for ctr_dir in sorted(glob.glob(os.path.join(d, "*"))):
        for f in glob.glob(os.path.join(ctr_dir, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f, newline="")))
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            for r in rows:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    src, out = sys.argv[1], sys.argv[2]
    res = {"correction": "gfx950: bytes read = FETCH_SIZE KB x 1024 x 2 for wide coalesced reads (>= 128 B per request), x 1 for 64-byte requests; bytes written = WRITE_SIZE KB x 1024 (MI355X_MICROARCH.md, HBM)",
           "collected_by": "tools/pmc_traffic.sh (one rocprofv3 --pmc pass per counter and program)"}
    k1 = load(os.path.join(src, "k1"))
    if "fp_fft_tile" in k1:
        f, w = k1["fp_fft_tile"]["FETCH_SIZE"], k1["fp_fft_tile"]["WRITE_SIZE"]
        # launches alternate pass A (64-byte segments: 4 consecutive 16-byte columns per stride -> 1:1, + the inter-pass
        # twiddle table) and pass B (contiguous rows: wide -> x2); told apart by their FETCH_SIZE
        big = [x for x in f if x > 1.2e7]
        small = [x for x in f if x <= 1.2e7]
        rd = (mean(big) * 1024 + mean(small) * 1024 * 2) / 2 if big and small else None
        wr = mean(w) * 1024
        res["fp_fft_tile"] = {"launches_sampled": len(f), "FETCH_SIZE_KB_pass_A": mean(big), "FETCH_SIZE_KB_pass_B": mean(small), "WRITE_SIZE_KB": mean(w),
                              "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": (rd + wr) if rd else None,
                              "algorithmic_bytes_per_launch": 2.0 * 1024 * (1 << 20) * 16,
                              "SQ_INSTS_VALU_per_launch": mean(k1["fp_fft_tile"].get("SQ_INSTS_VALU", [])),
                              "GRBM_GUI_ACTIVE_per_launch": mean(k1["fp_fft_tile"].get("GRBM_GUI_ACTIVE", []))}
    for prog, key in (("slig", "slig"), ("lch", "lch"), ("zk32", "zk32"), ("zksig", "zksig")):
        acc = load(os.path.join(src, prog))
        per = {}
        for k, cs in acc.items():
            if not any(x in k for x in ("bs_", "column_leaves", "merkle", "sc_grid", "grid256", "bindg", "eval_quad", "qw_scatter", "sumcheck_", "hquad", "dense_bind", "partials256", "bind256")):
                continue
            f, w = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
            per[k] = {"dispatches": max(len(f), len(w)), "FETCH_SIZE_KB_sum": sum(f), "WRITE_SIZE_KB_sum": sum(w), "FETCH_SIZE_KB_mean": mean(f), "WRITE_SIZE_KB_mean": mean(w),
                      "SQ_INSTS_VALU_mean": mean(cs.get("SQ_INSTS_VALU", [])), "GRBM_GUI_ACTIVE_mean": mean(cs.get("GRBM_GUI_ACTIVE", []))}
            if f and w:  # every one of these kernels reads whole 128-byte lines / 16 B per lane coalesced: wide
                per[k]["hbm_bytes_per_launch"] = (mean(f) * 2 + mean(w)) * 1024
        res["kernels_" + key] = per
    sl = res.get("kernels_slig", {})
    bs = {k: v for k, v in sl.items() if k.startswith("bs_")}
    if bs:
        # tools/slig_probe.py 20 1024 runs the encode TWICE (first call + timed call): per encode = half of the sums
        tot = sum((v["FETCH_SIZE_KB_sum"] * 2 + v["WRITE_SIZE_KB_sum"]) * 1024 for v in bs.values()) / 2
        rows, be = 1024, 1 << 20
        block = (be + 1) // 6
        res["slig_rs_encode"] = {"hbm_bytes_per_encode": tot, "launches_per_encode": sum(v["dispatches"] for v in bs.values()) / 2,
                                 "algorithmic_bytes_per_encode": rows * (block + be) * 16.0,
                                 "traffic_over_algorithmic": tot / (rows * (block + be) * 16.0)}
    if "column_leaves_kernel" in sl:
        res["column_leaves_kernel"] = dict(sl["column_leaves_kernel"])
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("kernels_")}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
