"""Dispatches of ONE steady-state proof, by kernel, from a rocprofv3 kernel trace of tools/bench_zk.py:
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/bench_zk.py 32 3
  python tools/dispatch_inventory.py OUT/*/*_kernel_trace.csv
A proof = the dispatches between the last two column_leaves_kernel launches (one commit + prove cycle)."""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void column_leaves_kernel")]
seg = rows[idx[-2]:idx[-1]]
name = lambda r: r["Kernel_Name"].split("(")[0]
cnt = collections.Counter(name(r) for r in seg)
dur = collections.Counter()
for r in seg:
    dur[name(r)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(json.dumps({"dispatches_per_proof": len(seg), "span_us": round((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3, 1),
                  "by_kernel": {k: {"calls": v, "us": round(dur[k] / 1e3, 1)} for k, v in cnt.most_common()}}, indent=1))
