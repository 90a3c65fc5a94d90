# One GPU call for both laboratories (run through gpurun from the repo root after `make -C mlx-audio-swift_amd/csrc && make -C tools/gemm_lab`):
#   gpurun --timeout 120 -- 'bash tools/gemm_lab/run_labs.sh'
# Results: gpurun_out/gemm_lab_rows32.jsonl, gemm_lab_rows16.jsonl, big_lab.jsonl (+ .err).  Each binary runs for a few seconds.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 40 tools/gemm_lab/gemm_lab 32 64 > gpurun_out/gemm_lab_rows32.jsonl 2> gpurun_out/gemm_lab_rows32.err
timeout 40 tools/gemm_lab/gemm_lab 16 64 > gpurun_out/gemm_lab_rows16.jsonl 2> gpurun_out/gemm_lab_rows16.err
timeout 60 tools/gemm_lab/big_lab 12000 20 > gpurun_out/big_lab.jsonl 2> gpurun_out/big_lab.err
python3 - <<'PY'
import json
for f in ("gemm_lab_rows32", "gemm_lab_rows16", "big_lab"):
    try: rows = [json.loads(l) for l in open(f"gpurun_out/{f}.jsonl") if l.startswith("{")]
    except OSError: rows = []
    best = {}
    for r in rows:
        k = r["shape"]
        if k not in best or r["us"] < best[k]["us"]: best[k] = r
    base = {r["shape"]: r for r in rows if r["variant"].startswith("product")}
    print(f, len(rows), "lines")
    for k, r in best.items():
        print(f"  {k:12s} product {base[k]['us']:8.2f} us   best {r['us']:8.2f} us  ({r['variant']}, max_rel {r['max_rel_vs_product']})")
PY
tail -3 gpurun_out/*lab*.err
