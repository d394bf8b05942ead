"""Turn rocprofv3 CSV output (kernel stats / counter collection) into the markdown summaries kept under profiles/."""
import collections, csv, glob, json, os, sys

def stats(prof_dir, out_md, title):
    f = glob.glob(os.path.join(prof_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out_md, "w") as o:
        o.write(f"# {title}\n\n| kernel | calls | total ms | avg us | % GPU time |\n|---|---|---|---|---|\n")
        for r in rows[:22]:
            o.write(f"| `{r['Name'][:96]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                    f"{float(r['AverageNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.2f} |\n")
        o.write(f"\ntotal GPU kernel time {tot/1e6:.1f} ms\n")

def pmc(prof_dir, key="gemm"):
    f = glob.glob(os.path.join(prof_dir, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = (r["Kernel_Name"][:80], r["Grid_Size"])
    return agg, names

if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        agg, names = pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "gemm")
        for d in sorted(agg, key=int):
            print(d, names[d], {k: f"{v:.5g}" for k, v in agg[d].items()})
