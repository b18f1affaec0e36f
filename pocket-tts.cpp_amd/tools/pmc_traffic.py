#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, one run each).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 bytes per
lane) coalesced reads at 64 bytes, so the read side of kernels whose traffic is such streams is doubled here
(`fetch_x2`); both the raw and the corrected figure are written.
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            if name.endswith(")") and "(" in name:
                name = name[:name.rindex("(")]      # drop the argument list, keep "(anonymous namespace)"
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: (tot[k] / cnt[k] * 1024.0, cnt[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fb, n = fetch.get(k, (0.0, 0))
        wbytes, _ = write.get(k, (0.0, 0))
        out[k] = dict(launches=n, fetch_raw=round(fb), fetch_x2=round(2 * fb), write=round(wbytes), traffic=round(2 * fb + wbytes))
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic"] * kv[1]["launches"])[:12]:
        print(f"{k[:70]:70s} n={v['launches']:6d} fetch_x2={v['fetch_x2']:>12d} write={v['write']:>12d}")


if __name__ == "__main__":
    main()
