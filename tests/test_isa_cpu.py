"""CPU tier (no GPU): properties of the COMPILED gfx950 code that the hand-scheduled kernels rely on and that a compiler change could take
away silently - hipcc cross-compiles here.

* `k_attn_decode2` loads its K/V tiles with asm statements and counted waits that hipcc does not track: the static audit of
  tools/isa_audit_attn2.py (no register of an in-flight load touched, no spill, no control flow with loads outstanding) must stay clean
  for every instantiation (one to four split-K slabs).
* The weight-streaming GEMMs run one or two waves per SIMD by design (R = 4: 160-230 registers): none of them, nor the glue and the
  256 x 256 Whisper GEMM, may spill (scratch traffic sits in the same vmcnt queue as the weight stream).
* The quantised R = 4 instantiations stay under 256 registers without scratch.
* The Whisper cross-attention schedule (`k_attn_decode<64, 2, XS>`) is plain C++ with compiler loads; what makes it fast is WHERE hipcc puts
  its waits - a pair of key tiles is requested before the wait for the pair in front of it (`s_waitcnt vmcnt(16)` behind sixteen requests,
  twice per tile count).  A loop form of the same code lost exactly that (vmcnt(0) at the loop header), so the property is pinned here.
One compile per source file (lm_kernels.hip ~40 s, the others ~15 s each)."""
import functools
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mlx-audio-swift_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@functools.lru_cache(maxsize=None)
def _compiled(src):
    """(assembly text, {mangled kernel name: {"vgprs": n, "scratch": bytes}}) of one source file: hipcc -S with
    -Rpass-analysis=kernel-resource-usage, once per file and test session."""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", asm,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(asm).read()
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        m = re.search(r"remark:\s+VGPRs: (\d+)", line)
        if m:
            cur["vgprs"] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m:
            cur["scratch"] = int(m.group(1))
    return text, out


def _resource_usage(src):
    return _compiled(src)[1]


def test_attention_asm_schedule_audit_is_clean():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit_attn2.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert r.stdout.count("0 violations") >= 4 and "violations: 0" in r.stdout        # NS = 1..4


def test_step_chain_kernels_do_not_spill():
    use = _resource_usage("lm_kernels.hip")
    gemm = {k: v for k, v in use.items() if "k_gemm_skinny" in k}
    assert len(gemm) >= 60                                              # MT 1..4 x the instantiated (R, epilogue, KSB, U) set
    for k, v in gemm.items():
        assert v["scratch"] == 0 and v["vgprs"] <= 256, (k, v)
    r4 = [v["vgprs"] for k, v in gemm.items() if re.search(r"k_gemm_skinnyILi2ELi4E", k)]
    assert r4 and max(r4) <= 232                                        # four n-tiles per wave at 32 rows: no pressure on the 256-register budget
    for name in ("k_glue_cpt", "k_glue4", "k_attn_decode2", "k_embed_rmsnorm"):
        hit = {k: v for k, v in use.items() if name in k}
        assert hit, name
        for k, v in hit.items():
            assert v["scratch"] == 0, (k, v)


def test_whisper_256_tile_gemm_and_quantised_r4_gemm_do_not_spill():
    use = _resource_usage("whisper_kernels.hip")
    big3 = {k: v for k, v in use.items() if "k_gemm_big3" in k}
    assert len(big3) == 8                                               # four epilogues x two wave grids
    for k, v in big3.items():
        assert v["scratch"] == 0 and v["vgprs"] <= 256, (k, v)
    useq = _resource_usage("lm_qgemm.hip")
    q4 = {k: v for k, v in useq.items() if re.search(r"k_gemm_skinny_qILi[12]ELi4E", k)}
    assert len(q4) >= 24
    for k, v in q4.items():
        assert v["scratch"] == 0 and v["vgprs"] <= 256, (k, v)


def test_whisper_cross_attention_keeps_two_pairs_of_key_tiles_in_flight():
    """k_attn_decode<64, 2, true, QP>: the plain cross-attention schedule (QP = 0) and the one with LayerNorm 2 + the query projection in
    its prologue (QP = 4 / 8 slabs per trip, round 6) - no scratch, and on every tile-count path two pairs of K/V tiles in flight."""
    text, use = _compiled("lm_kernels.hip")
    names = sorted(k for k in use if re.search(r"k_attn_decodeILi64ELi2ELb1ELi[048]E", k))
    assert len(names) == 3, names
    for name in names:
        i = text.index("\n" + name + ":")
        body = text[i: text.index(".Lfunc_end", i)]
        ins = [l.strip() for l in body.split("\n") if re.match(r"^\s+[a-z_0-9]+\b", l)]
        # no vector register spilled and no scratch instruction anywhere; the projection variants park four SCALAR registers (their seven
        # extra pointers) in vector lanes - a 20-byte frame on paper, no memory traffic
        assert use[name]["vgprs"] <= 256 and not any("scratch_" in l for l in ins), (name, use[name])
        assert use[name]["scratch"] == 0 or (not name.endswith("Li0EEv10AttnParams") and use[name]["scratch"] <= 32), (name, use[name])
        # K/V requests (non-temporal 16-byte loads) since the previous vmcnt wait, at every vmcnt wait (address arithmetic sits between the loads)
        waits, n = [], 0
        for l in ins:
            if l.startswith("global_load_dwordx4") and l.endswith(" nt"):
                n += 1
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                waits.append((n, int(re.search(r"vmcnt\((\d+)\)", l).group(1))))
                n = 0
        assert sum(k for k, _ in waits) + n == 16 + 4 * 16, (name, waits)    # the first group up front (8 + 8), two pairs on either tile-count path
        # a wait that follows a full pair of requests (sixteen loads or more since the last wait) must leave that pair outstanding: the pair is
        # requested BEFORE the wait for the group in front of it - twice on the six-tile path, twice on the five-tile path
        behind_pair = [w for k, w in waits if k >= 16]
        assert len(behind_pair) == 4 and all(w >= 16 for w in behind_pair), (name, waits)
        if not name.endswith("Li0EEv10AttnParams"):
            # the projection prologue: its waits in front of the first MFMA count loads (the first K/V tile stays in flight behind them)
            first_mfma = next(j for j, l in enumerate(ins) if l.startswith("v_mfma"))
            pro = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in ins[:first_mfma] if l.startswith("s_waitcnt") and "vmcnt" in l]
            assert pro and min(pro) >= 8, (name, pro)


def test_token_engine_keeps_its_tile_buffers_in_registers():
    """csrc/token_engine.hip holds a phase's weight tiles in registers while the previous phase is still running (gate|up: 176 registers
    per matrix wave).  Two compiler behaviours once put them into scratch - tile addresses hoisted out of the layer loop (~200 registers
    of addresses) and the output projection's tiles live across the whole loop; the source works around both (opaque scalar ids, rotated
    loop).  Pinned here: no scratch at all on 2 / 4 / 8 XCDs (the product's default is 4), a bounded remainder on 1 XCD (ten tile rows of
    gate|up per worker), and the barrier the two programs meet at must not drain vector-memory loads."""
    text, use = _compiled("token_engine.hip")
    eng = {k: v for k, v in use.items() if "k_token_engine" in k}
    assert len(eng) == 4, list(eng)
    for k, v in eng.items():
        one = "ILi1E" in k
        assert v["vgprs"] <= 256, (k, v)
        assert v["scratch"] <= (256 if one else 0), (k, v)
    # te_sync() = s_waitcnt lgkmcnt(0) + s_barrier: the instruction in front of (almost) every barrier is the lgkmcnt-only wait
    name = [k for k in eng if "ILi4E" in k][0]
    i = text.index("\n" + name + ":")
    body = [l.strip() for l in text[i: text.index(".Lfunc_end", i)].split("\n")]
    waits = [body[j - 1] for j, l in enumerate(body) if l.startswith("s_barrier") and j > 0]
    assert len(waits) >= 40
    assert sum(1 for w_ in waits if w_.startswith("s_waitcnt") and "vmcnt" not in w_) >= 0.9 * len(waits), waits[:8]
