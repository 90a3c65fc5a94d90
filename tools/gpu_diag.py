"""Run every GPU parity check independently (each pytest node in its own process with a timeout), so one
crashing kernel cannot hide the others.  Writes gpurun_out/diag.json + per-test logs.
    python tools/gpu_diag.py [pattern]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q"], cwd=ROOT,
                       capture_output=True, text=True)
    nodes = [l.strip() for l in r.stdout.splitlines() if "::" in l and pat in l]
    res = {}
    for n in nodes:
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, "-m", "pytest", n, "-x", "-q", "--no-header", "-p", "no:cacheprovider"],
                               cwd=ROOT, capture_output=True, text=True, timeout=420)
            status = "pass" if p.returncode == 0 else "FAIL"
            tail = (p.stdout + p.stderr)[-3500:]
        except subprocess.TimeoutExpired as e:
            status, tail = "TIMEOUT", str(e)[-500:]
        res[n] = {"status": status, "sec": round(time.time() - t0, 1)}
        print(f"[{status}] {n} ({res[n]['sec']} s)", flush=True)
        if status != "pass":
            print(tail, flush=True)
            with open(os.path.join(OUT, "diag_" + n.replace("/", "_").replace("::", "__").replace("[", "_").replace("]", "") + ".log"), "w") as f:
                f.write(tail)
    with open(os.path.join(OUT, "diag.json"), "w") as f:
        json.dump(res, f, indent=1)
    bad = [n for n, v in res.items() if v["status"] != "pass"]
    print(f"{len(res) - len(bad)}/{len(res)} passed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
