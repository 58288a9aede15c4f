"""profiles/rNN_sustained.json from two bench.py runs on the GPU box:
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > line.json
  LT_BENCH_TREND=32 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-hydro-timing 2> trend.err
usage: python3 profiles/tools/sustained_json.py line.json trend.err > profiles/rNN_sustained.json"""
import json, re, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
blocks, cur = [], []
for ln in open(sys.argv[2]):
    m = re.match(r"sustained steps (\d+)\.\.(\d+): ([0-9.]+) ms per step", ln)
    if m:
        if int(m.group(1)) == 1 and cur:
            blocks.append(cur); cur = []
        cur.append(float(m.group(3)))
if cur:
    blocks.append(cur)
out = {
    "command": "python3 bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command): the sustained series runs on the same engine right after the 20-step window",
    "kernel_source_hash": c.get("kernel_source_hash"),
    "window": {"value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"]},
    "sustained": {k: c[k] for k in c if k.startswith("sustained_")},
    "turbulence_window": {"value": c.get("turb_value"), "ms_per_step": c.get("turb_ms_per_step")},
    "turbulence_sustained": {k: c[k] for k in c if k.startswith("turb_sustained_")},
    "per_32_step_blocks_ms": {
        "note": "LT_BENCH_TREND=32 with the same command (a fence every 32 steps, another run on the same box): the synthetic case's work per particle drifts",
        "advection": blocks[0] if blocks else None,
        "turbulence": blocks[1] if len(blocks) > 1 else None,
    },
}
print(json.dumps(out, indent=1))
