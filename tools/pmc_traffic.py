#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of `bench.py --no-graph` into
profiles/pmc_traffic.json: HBM bytes per launch for every kernel symbol, keyed the way bench.py names kernels.

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> profiles/pmc_traffic.json [forwards]

forwards: the number of 16-pair forwards the profiled command issued (timed + warm-up steps + bench.py's eager measurement
passes: --steps 2 --warmup 1 -> 3 + 2 = 5); the blocker GEMMs of the measurement passes (the 256x256-tile gemm8 instance on
8192^3) are taken out, and "_step" = bytes of one 16-pair forward.

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE under-counts wide coalesced reads by 2x,
so bytes = (2 * FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024.  The first launches of every symbol (warm-up) are included;
shapes repeat every step so the per-launch mean is over identical launch sets."""
import collections
import csv
import json
import re
import sys


def bench_key(name):
    m = re.search(r"(gemm8_kernel<[^>]*>)", name)          # demangled template instance: exactly bench.py's key
    if m:
        return m.group(1)
    m = re.search(r"match_kernel<(true|false)>", name) or re.search(r"match_kernelILb([01])E", name)
    if m:
        return "match_kernel" + ("+scores" if m.group(1) in ("true", "1") else "")
    for k in ("sra_q_kernel", "sra_block_kernel", "sra_kernel"):
        if k in name:
            return k
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELb([01])ELi(\d+)ELb([01])EE", name)
    if m:
        return "gemm_kernel<%s,%s,%s,%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2), m.group(3),
                                             "conv" if m.group(4) == "1" else "dense")
    m = re.search(r"attn_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)EE", name)
    if m:
        return "attn_kernel<%s,%s,%s,%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2), m.group(3),
                                             m.group(4))      # bench.py appends "+scores" for the correlation launch
    m = re.search(r"N_1\d+(\w+?)_kernelI", name) or re.search(r"(\w+)_kernel", name)
    return ("emip_" + m.group(1)) if m else name[:60]


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(cnt[k])) for k in tot}


def main():
    fetch, write, out = sys.argv[1:4]
    forwards = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    fe, wr = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        key = bench_key(k)
        e = res.setdefault(key, {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "launches_profiled": 0, "symbols": []})
        e["FETCH_SIZE_KB"] += f
        e["WRITE_SIZE_KB"] += w
        e["launches_profiled"] += n
        e["symbols"].append(k[:120])
    for e in res.values():
        n = e["launches_profiled"]
        e["FETCH_SIZE_KB_per_launch"] = round(e.pop("FETCH_SIZE_KB") / n, 1)
        e["WRITE_SIZE_KB_per_launch"] = round(e.pop("WRITE_SIZE_KB") / n, 1)
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"]) * 1024)
        e["note"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of wide coalesced reads"
    total = sum(e["hbm_bytes_per_launch"] * e["launches_profiled"] for e in res.values())
    res["_total"] = {"hbm_bytes_all_profiled_launches": int(total), "note": "divide by the forwards profiled (warm-up included)"}
    if forwards:
        blockers = sum(e["hbm_bytes_per_launch"] * e["launches_profiled"] for k, e in res.items()
                       if k.startswith("gemm8_kernel<256, 256") and ", false, false, false, false>" in k and not k.startswith("_"))
        res["_step"] = {"hbm_bytes_per_16pair_step": int((total - blockers) / forwards), "forwards_profiled": forwards,
                        "blocker_bytes_removed": int(blockers),
                        "note": "(_total - blocker GEMMs of bench.py's measurement passes) / forwards"}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
