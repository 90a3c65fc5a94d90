"""Decode attention (`k_attn_decode2`) against the context length: one process per context (`MIS_TIME_ATTN_CTX` is read once per process),
an Orpheus-width model of `--layers` layers with synthetic weights (the probe rotates over the layers' caches, so that the K/V stream of
consecutive launches cannot come out of the 256 MB Infinity Cache: layers x 2 x 32 rows x 8 kv heads x context x 128 x 2 B must exceed it).
What it is for: a wave owns the key tiles wave, wave + 8, ... (32 keys each), so the launch lasts as long as its fullest wave - the step
from 256 to 288 keys (one wave gets a second tile) against the slope between the steps says what a balanced distribution could gain.
Usage: python tools/attn_ctx_sweep.py out.json [--layers 16] ctx1 ctx2 ...   (one JSON object per context on stdout)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) >= 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import mlx_audio_swift_amd as mas
    layers, rows, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    cfg = mas.LlamaTTSConfiguration(hidden_size=3072, num_hidden_layers=layers, intermediate_size=8192, num_attention_heads=24,
                                    num_key_value_heads=8, head_dim=128, vocab_size=156940, rope_theta=500000.0)
    lm = mas.LlamaTTSModel.synthetic(cfg, seed=4321)
    lm.time_gemm(5, rows, iters=iters)                                          # first series of a process: clocks
    best = None
    for _ in range(3):
        ms, by = lm.time_gemm(5, rows, iters=iters)
        best = ms if best is None else min(best, ms)
    print(json.dumps({"us": round(best * 1e3, 3), "bytes": by, "GBps": round(by / best / 1e6, 1)}))
    sys.exit(0)

out_path = sys.argv[1]
args = sys.argv[2:]
layers = 16
if args and args[0] == "--layers":
    layers = int(args[1]); args = args[2:]
res = []
for ctx in [int(a) for a in args]:
    e = dict(os.environ, MIS_TIME_ATTN_CTX=str(ctx))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(layers), "32", "112"], env=e, capture_output=True, text=True)
    try:
        row = dict(json.loads(r.stdout.strip().splitlines()[-1]), context=ctx, key_tiles=(ctx + 1 + 31) // 32, layers=layers,
                   tiles_of_fullest_wave=((ctx + 1 + 31) // 32 + 7) // 8)
    except Exception as ex:
        row = {"context": ctx, "error": repr(ex), "stderr": r.stderr[-400:]}
    res.append(row)
    print(json.dumps(row), flush=True)
    json.dump(res, open(out_path, "w"), indent=1)
