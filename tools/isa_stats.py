"""Static ISA statistics of every kernel in the library (no GPU needed): instruction count, MFMA / VMEM / LDS / barrier counts,
VGPRs, AGPRs, LDS bytes, occupancy - from `hipcc -S` and the kernel-resource-usage remarks.  For a kernel that executes its body
once per launch (everything on the decode step chain except the GEMM / attention main loops) the instruction count times 4 cycles
per wave64 VALU issue is a lower bound of its duration; DESIGN.md ("Instruction count is part of the per-kernel floor").
Usage: python tools/isa_stats.py [out.json]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mlx-audio-swift_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
        got = out.strip().split("\n")
        return got if len(got) == len(names) else names
    except OSError:
        return names


def stats_of(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, *FLAGS, "-S", path, "-o", asm, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        txt = open(asm).read()
    res = {}
    cur = None
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds_bytes", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    for name in re.findall(r"^(_Z\w+):\s*; @", txt, re.M):
        i = txt.index("\n" + name + ":")
        body = txt[i: txt.index(".Lfunc_end", i)]                     # (early returns put s_endpgm in the middle of a body)
        ins = [l.strip().split()[0] for l in body.split("\n") if re.match(r"^\s+[a-z_0-9]+\b", l) and not l.strip().startswith(";")]
        ins = [x for x in ins if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", x)]
        d = res.setdefault(name, {})
        d.update(instructions=len(ins), mfma=sum(x.startswith("v_mfma") for x in ins),
                 vmem=sum(x.startswith(("global_", "buffer_", "flat_")) for x in ins), lds=sum(x.startswith("ds_") for x in ins),
                 barriers=sum(x == "s_barrier" for x in ins), valu=sum(x.startswith("v_") and not x.startswith("v_mfma") for x in ins),
                 salu=sum(x.startswith("s_") for x in ins))
    return res


def main():
    out = {}
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        st = stats_of(os.path.join(CSRC, f))
        names = list(st)
        for n, dn in zip(names, demangle(names)):
            if "instructions" in st[n]:
                out[f + " :: " + dn[:110]] = st[n]
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_isa_stats.json")
    json.dump(out, open(path, "w"), indent=1)
    hot = ("k_attn_decode<128, 2>", "k_reduce_residual_rmsnorm", "k_embed_rmsnorm", "k_samp_", "k_gemm_skinny<2, 2, 2, 4, false>",
           "k_gemm_skinny<2, 2, 0, 4, false>", "k_gemm_skinny<2, 2, 1, 1, false>")
    for k, v in out.items():
        if any(h in k for h in hot):
            print(f"{k.split(' :: ')[1][:60]:62s} instr {v['instructions']:5d}  valu {v['valu']:5d}  mfma {v['mfma']:4d}  vmem {v['vmem']:4d}  "
                  f"lds {v['lds']:4d}  vgpr {v.get('vgprs', -1):4d}  occ {v.get('occupancy', -1)}")
    print("wrote", path, len(out), "kernels")


if __name__ == "__main__":
    main()
