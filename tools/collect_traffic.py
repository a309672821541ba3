#!/usr/bin/env python3
"""Reduce two `rocprofv3 --pmc` passes over bench.py (FETCH_SIZE, then WRITE_SIZE -- they cannot share a
pass on gfx950: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2) to HBM-side bytes per launch per
kernel, with the corrections /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream => x2; WRITE_SIZE is exact for
16-B-per-lane streaming stores; both are in KiB.

    python tools/collect_traffic.py <dir with pmc_fetch/ and pmc_write/> profiles/r01_traffic.json
(the passes themselves are run by tools/profile_bench.sh on the GPU box)"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(pattern, counter):
    agg = collections.defaultdict(list)
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_(?:layer|chain|gather|propagate|linear_split|linear|wide|mlp2r|mlp2)<[^>]*>)", r["Kernel_Name"])
            if m:
                agg[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fetch, nf = per_kernel(f"{src}/pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE")
    write, nw = per_kernel(f"{src}/pmc_write/**/*counter_collection.csv", "WRITE_SIZE")
    out = {"_note": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; gfx950 correction "
                    "per MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read)",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                             "launches_sampled": [nf.get(k, 0), nw.get(k, 0)],
                             "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    # the 8-member edge-MLP block (its own pair of passes: pmc8_fetch / pmc8_write), keyed with an @8members suffix
    fetch8, nf8 = per_kernel(f"{src}/pmc8_fetch/**/*counter_collection.csv", "FETCH_SIZE")
    write8, nw8 = per_kernel(f"{src}/pmc8_write/**/*counter_collection.csv", "WRITE_SIZE")
    for k in sorted(set(fetch8) & set(write8)):
        if not k.startswith("k_mlp2"):
            continue
        f, w = fetch8[k], write8[k]
        out["kernels"][k + "@8members"] = {"FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                                           "launches_sampled": [nf8.get(k, 0), nw8.get(k, 0)],
                                           "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
