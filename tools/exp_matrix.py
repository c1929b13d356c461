"""Runs bench.py (headline step only) under a list of environment settings, one subprocess each, and prints ms/step.
Development aid for any environment knob:  python tools/exp_matrix.py OUT.json NAME=K1:V1,K2:V2 ..."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1]
extra = os.environ.get("EXP_BENCH_ARGS", "--steps 5 --warmup 2 --no-cpu-baseline --no-dense --no-roofline --no-parity --no-secondary").split()
rows = []
for spec in sys.argv[2:]:
    name, _, kv = spec.partition("=")
    env = dict(os.environ)
    for item in filter(None, kv.split(",")):
        k, _, v = item.partition(":")
        env[k] = v
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                           text=True, timeout=int(os.environ.get("EXP_RUN_TIMEOUT", "120")))
        line = next((l for l in reversed(r.stdout.strip().splitlines()) if l.startswith("{")), None)
    except subprocess.TimeoutExpired:
        r, line = None, None
    row = {"name": name, "env": kv, "wall_s": round(time.time() - t0, 1)}
    if line:
        d = json.loads(line)
        row.update(ms_per_step=d["ms_per_step"], value=d["value"], final_cost=d.get("final_cost"))
    else:
        row.update(error=(r.stderr or r.stdout)[-400:] if r is not None else "timeout")
    rows.append(row)
    print(json.dumps(row), flush=True)
    json.dump(rows, open(out_path, "w"), indent=1)
