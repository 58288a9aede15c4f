#!/usr/bin/env python3
"""Reduce the rocprofv3 csv files of profiles/tools/collect.sh to a short summary.

kernel_stats.csv : the --stats table of the timed run (unchanged)
pmc.json         : per kernel, mean counter value per launch for every counter
                   collected, plus HBM-side bytes per launch:
                   fetch_bytes = FETCH_SIZE [KiB -> B] x 2, write_bytes = WRITE_SIZE [KiB -> B].
                   The factors are measured, not assumed: profiles/r02_calib/calib.txt (kernels of
                   known traffic, profiles/tools/calibrate.sh) shows FETCH_SIZE x 1024 = exactly half
                   the bytes of the 128-byte lines touched for EVERY read shape the engine uses (wide
                   16 B/lane, coalesced 8 B, 4 / 8 / 16 / 32-byte gathers), and WRITE_SIZE x 1024 = the
                   bytes written for coalesced stores (3.6x the payload for per-lane runs of 32-byte
                   records, which is why no kernel writes that way).
                   "_meta": the hash of the kernel sources the library was built from (bench.py only
                   quotes these numbers for a library with the same hash).
summary.txt      : the same, readable
"""
import csv, glob, json, os, sys, collections

def main(out):
    sm = os.path.join(out, "summary")
    os.makedirs(sm, exist_ok=True)
    stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
    lines = []
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(sm, "kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(rows)
        lines.append("kernel durations (rocprofv3 --kernel-trace --stats)")
        # (rocBLAS' Cijk_* products are bench.py's clock pre-warm -- some tens of milliseconds of fp32 matrix products before the warm-up
        # steps of each timed series, outside the timed region: kept in kernel_stats.csv, not listed here)
        warm = [r for r in rows[1:] if r[0].startswith("Cijk_")]
        for r in [r for r in rows[1:] if not r[0].startswith("Cijk_")][:11]:
            lines.append("  %-58s calls %4s  avg %10.1f us  %5s %%" % (r[0][:58], r[1], float(r[3]) / 1e3, r[4][:5]))
        if warm:
            lines.append("  (percentages of all GPU time, of which %.1f %% is bench.py's clock pre-warm: %d rocBLAS products, not listed)"
                         % (sum(float(r[4]) for r in warm), sum(int(r[1]) for r in warm)))
    for name in ("stats", "pmc_sq", "pmc_fetch", "pmc_write"):
        p = os.path.join(out, name + ".bench.json")
        if os.path.exists(p) and os.path.getsize(p) and name == "stats":
            open(os.path.join(sm, "bench.json"), "w").write(open(p).read())
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for p in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(float)  # (dispatch, kernel, counter) -> summed over instances
            for r in csv.DictReader(open(p)):
                per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (d, k, c), v in per.items():
                acc[k][c].append(v)
    res = {}
    for k, cs in acc.items():
        if not (k.startswith("ltd::") or k.startswith("void ltd::") or k.startswith("ltmt::")):
            continue
        e = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in e:
            e["fetch_bytes_per_launch"] = e["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0
        if "WRITE_SIZE" in e:
            e["write_bytes_per_launch"] = e["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        res[k] = e
    meta = {}
    for name in ("stats", "pmc_sq", "pmc_fetch", "pmc_write"):
        p = os.path.join(out, name + ".bench.json")
        try:
            meta["source_hash"] = json.loads(open(p).read().strip().splitlines()[-1])["config"]["kernel_source_hash"]
            break
        except Exception:
            pass
    res["_meta"] = meta
    json.dump(res, open(os.path.join(sm, "pmc.json"), "w"), indent=1, sort_keys=True)
    res.pop("_meta")
    lines.append("")
    lines.append("counters, mean per launch")
    for k in sorted(res, key=lambda k: -res[k].get("SQ_BUSY_CYCLES", {"mean_per_launch": 0})["mean_per_launch"]):
        e = res[k]
        g = lambda c: e.get(c, {"mean_per_launch": float("nan")})["mean_per_launch"]
        waves = g("SQ_WAVES")
        lines.append("  %s" % k[:70])
        lines.append("    waves %.0f  VALU instr/wave %.0f  SALU instr/wave %.0f" % (waves, g("SQ_INSTS_VALU") / waves, g("SQ_INSTS_SALU") / waves))
        wc = g("SQ_WAVE_CYCLES")
        lines.append("    of wave cycles: waiting %.1f %%  issue-stalled %.1f %%  VALU active %.1f %%" % (
            100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc))
        lines.append("    HBM-side bytes: read %.1f MB  written %.1f MB" % (
            e.get("fetch_bytes_per_launch", float("nan")) / 1e6, e.get("write_bytes_per_launch", float("nan")) / 1e6))
    open(os.path.join(sm, "summary.txt"), "w").write("\n".join(lines) + "\n")

if __name__ == "__main__":
    main(sys.argv[1])
