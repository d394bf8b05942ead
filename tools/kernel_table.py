"""Summarise a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) as a markdown table: python tools/kernel_table.py FILE [title]."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {title}\n\n| kernel | calls | total ms | avg us | % GPU time |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{100 * float(r['TotalDurationNs']) / tot:.2f} |")
print(f"\ntotal GPU kernel time {tot / 1e6:.1f} ms")
