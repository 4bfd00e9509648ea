#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats / PMC counter collection) into small files for profiles/."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def pmc_summary(csv_path):
    rows = list(csv.DictReader(open(csv_path)))
    agg = defaultdict(lambda: defaultdict(list))
    for r in rows:
        k = r.get("Kernel_Name") or r.get("Kernel Name")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: dict(mean=sum(v) / len(v), n=len(v), min=min(v), max=max(v)) for c, v in cs.items()} for k, cs in agg.items()}


if __name__ == "__main__":
    out = {}
    for p in sys.argv[2:]:
        for f in Path(p).rglob("*counter_collection.csv"):
            for k, v in pmc_summary(f).items():
                out.setdefault(k, {}).update(v)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps({k[:60]: {c: round(x["mean"]) for c, x in v.items()} for k, v in out.items()}, indent=0)[:3000])
