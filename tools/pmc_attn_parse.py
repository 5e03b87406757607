"""Summarises tools/pmc_attn.sh output: per kernel name, the mean of every counter over its dispatches."""
import csv, glob, collections, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_attn"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "attn" not in name: continue
        agg[(re.search(r"attn_\w+<[^>]*>", name) or re.search(r"attn_\w+", name)).group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} {sum(v) / len(v):16.0f}  (n={len(v)})")
