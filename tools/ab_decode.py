"""A/B of decode-step variants on the headline workload: runs bench.py (2 timed generates, no cpu baseline) once per named
environment setting and prints / writes value, decode ms, step ms and the per-kernel probe timings of each.
Usage: python tools/ab_decode.py out.json name1:VAR=val+VAR2=val name2:... (variables separated by "+": values may hold commas;
the name "base" with no variables is always run first)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1]
variants = [("base", {})]
for spec in sys.argv[2:]:
    name, _, rest = spec.partition(":")
    env = dict(kv.split("=", 1) for kv in rest.split("+") if kv)
    variants.append((name, env))
rows = []
for name, env in variants:
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], env=e,
                       capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        row = {"name": name, "env": env, "value": round(j["value"], 2), "decode_ms": round(j["phases_ms"]["decode"], 2),
               "prefill_ms": round(j["phases_ms"]["prefill"], 2), "codec_ms": round(j["phases_ms"]["codec"], 2),
               "step_ms": round(j["roofline"]["step"]["ms"], 4), "frame_slot_step_ms": j["sampler"].get("frame_slot_step_ms"),
               "kernels_us": {k: v["us"] for k, v in j["roofline"]["kernels"].items()}}
    except Exception as ex:                                       # keep going: one broken variant must not lose the others
        row = {"name": name, "env": env, "error": repr(ex), "stderr": r.stderr[-600:], "stdout": r.stdout[-300:]}
    rows.append(row)
    print(json.dumps(row), flush=True)
    json.dump(rows, open(out_path, "w"), indent=1)
