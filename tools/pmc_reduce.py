"""Reduce the rocprofv3 --pmc passes of tools/pmc_traffic.sh to profiles/rNN_pmc/traffic.json.
  python tools/pmc_reduce.py <fetch_dir> <write_dir> <out_json>
Per kernel: mean FETCH_SIZE / WRITE_SIZE per dispatch (KiB).  gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
tallies 128-byte requests at 64 bytes for wide coalesced streaming reads -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is
uncalibrated in general and is taken at face value here because it reproduces the kernels' known output sizes exactly
(gate+up: 32 x 8192 bf16 = 512 KiB)."""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            acc.setdefault(k, []).append(float(row["Counter_Value"]))
    return acc


def sha():
    h = hashlib.sha1()
    for f in ("lm_kernels.hip", "lm_kernels.h", "lm_sampler.hip", "lm_engine.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "mlx-audio-swift_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    fd, wd, out = sys.argv[1:4]
    F, Wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    # algorithmic bytes per launch (Orpheus-3B, batch 32, mean context 368), see DESIGN.md section 3
    d, ff, H, Hkv, D, V = 3072, 8192, 24, 8, 128, 156940
    alg = {"gate_up": 2 * 2 * ff * d, "lm_head": 2 * V * d, "qkv": 2 * (H + 2 * Hkv) * D * d, "o_proj": 2 * d * H * D, "down": 2 * d * ff,
           "attn_decode_ctx368": 32 * 369 * 2 * Hkv * D * 2}
    res = {"_how": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python "
                   "tools/pmc_probe.py, separate passes; per-dispatch means; read bytes = 2 * FETCH_SIZE KiB * 1024 (gfx950 counts "
                   "128-B requests at 64 B), WRITE_SIZE at face value",
           "kernel_source_sha1": sha(), "kernels": {}}

    def pick(sub, exclude=()):
        ks = [k for k in F if sub in k and not any(e in k for e in exclude)]
        return ks

    table = {}
    for k in sorted(F):
        n = len(F[k])
        fe = sum(F[k]) / n
        wr = sum(Wr.get(k, [0.0])) / max(len(Wr.get(k, [0.0])), 1)
        table[k] = {"dispatches": n, "FETCH_SIZE_KiB": round(fe, 1), "WRITE_SIZE_KiB": round(wr, 1),
                    "hbm_bytes_per_launch": int(2 * fe * 1024 + wr * 1024)}
    res["kernels"] = table
    # named entries: the round-4 build gives every role its own instantiation k_gemm_skinny<MT, R, epilogue, KSB, U>
    for name, key in (("gate_up", "k_gemm_skinny<2, 4, 2, 4, 3>"), ("lm_head", "k_gemm_skinny<2, 4, 1, 4, 3>"), ("qkv", "k_gemm_skinny<2, 4, 0, 2, 2>"),
                      ("o_proj", "k_gemm_skinny<2, 2, 0, 4, 4>"), ("down", "k_gemm_skinny<2, 2, 0, 2, 4>"), ("attn_decode_ctx368", "k_attn_decode"),
                      ("reduce_residual_rmsnorm", "k_glue4")):
        ks = [k for k in table if key in k]
        if ks:
            e = dict(table[ks[0]]); e["kernel"] = ks[0]
            if name in alg:
                e["algorithmic_bytes_per_launch"] = alg[name]; e["ratio"] = round(e["hbm_bytes_per_launch"] / alg[name], 3)
                if name in ("qkv", "o_proj", "down"):
                    e["note"] = "traffic includes the f32 partial slabs written for the consumer (S x 32 x N x 4 B per launch)"
            res[name] = e
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("kernels", "_how")}, indent=1))


if __name__ == "__main__":
    main()
