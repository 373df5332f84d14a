#!/usr/bin/env python3
"""Instruction-class histogram of a kernel's INNERMOST loops, priced with the measured issue costs.

    python tools/isa_hist.py [--kernel SUBSTR] [--depth N] [--costs profiles/r04_valu_rates.txt] [raster.s]

Compiles gslam_amd/csrc/raster.hip to gfx950 assembly when no .s file is given (device only, the flags of csrc/build.py),
cuts out the kernel whose mangled name contains SUBSTR (default: the fused tracking rasteriser), and for every basic block
that LLVM's loop annotation places at loop depth >= N (default 2: the per-survivor loops) prints the instructions by class:

    fma    v_fma / v_fmac / v_mul / v_add / v_sub (f32)           full rate        2.4 cycles per SIMD and wave-instruction
    half   compares, v_cndmask, min / max / med3, integer / bit ops, v_mov, DPP forms, v_readlane, permlane swaps  4.2
    pk     v_pk_*_f32                                              4.3
    trans  v_exp / v_rcp / v_log / v_rsq / v_sqrt                  8.2
    mfma   v_mfma_*                                                issue slot only (8), the matrix pipe runs beside the VALU
    lds    ds_*                                                    no VALU issue time
    salu   s_* (scalar unit, own issue port), s_nop / s_waitcnt listed apart
    vmem   global_* / buffer_* / flat_*

The costs are the wall-clock figures of tools/ubench/valu_rates2.hip at 4-8 waves per SIMD (profiles/r04_valu_rates.txt); a
costs file given with --costs overrides the class defaults per mnemonic where it lists one.  The sum over a block is what one
trip of that block costs a SIMD in VALU issue time; the table is what VERDICT r03 item 1(a) asked for.
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASS_COST = {"fma": 2.4, "half": 4.2, "pk": 4.3, "trans": 8.2, "mfma": 8.0, "lds": 0.0, "salu": 0.0, "nop": 0.0,
              "wait": 0.0, "vmem": 0.0, "branch": 0.0}

FULL_RATE = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_fma_mix")
TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag")


def classify(mn: str, text: str) -> str:
    if mn.startswith("v_mfma"):
        return "mfma"
    if mn.startswith("v_pk_"):
        return "pk"
    if mn.startswith(TRANS):
        return "trans"
    if mn.startswith("v_"):
        dpp = ("row_" in text) or ("quad_perm" in text) or ("wave_" in text) or mn.endswith("_dpp")
        if mn.startswith(FULL_RATE) and not dpp:
            return "fma"
        return "half"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn == "s_nop":
        return "nop"
    if mn.startswith("s_waitcnt"):
        return "wait"
    if mn.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    return "salu"


def compile_asm() -> str:
    out = os.path.join(tempfile.gettempdir(), "gsx_raster_isa.s")
    cmd = ["hipcc", "-S", "--cuda-device-only", "-O3", "-std=c++17", "--offload-arch=gfx950", "-DNDEBUG",
           "-fno-slp-vectorize", f"-I{ROOT}/include", f"{ROOT}/gslam_amd/csrc/raster.hip", "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def kernel_text(asm: str, substr: str) -> tuple[str, list[str]]:
    lines = open(asm).read().splitlines()
    start = None
    name = ""
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\S+):", ln)
        if m and substr in m.group(1):
            start, name = i, m.group(1)
            break
    if start is None:
        raise SystemExit(f"no kernel matching {substr!r}")
    body = []
    for ln in lines[start + 1:]:
        body.append(ln)
        if "s_endpgm" in ln:
            break
    return name, body


def blocks(body: list[str]):
    """Yield (label, depth, [(mnemonic, text)]) per basic block; inline-asm bodies are ordinary lines of the stream."""
    label, depth, cur = "entry", 0, []
    for ln in body:
        m = re.match(r"^(\.LBB\S+):(.*)$", ln)
        m2 = re.match(r"^; %bb\.(\d+):(.*)$", ln)
        if m or m2:
            if cur:
                yield label, depth, cur
            label = m.group(1) if m else f"bb.{m2.group(1)}"
            tail = (m.group(2) if m else m2.group(2))
            d = re.search(r"Depth=(\d+)", tail)
            depth = int(d.group(1)) if d else (depth if m2 else 0)
            cur = []
            continue
        d = re.search(r"Loop.*Depth=(\d+)", ln)
        if d and ln.lstrip().startswith(";"):
            depth = max(depth, int(d.group(1)))
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        mn = t.split()[0]
        cur.append((mn, t))
    if cur:
        yield label, depth, cur


def load_costs(path: str | None) -> dict[str, float]:
    costs: dict[str, float] = {}
    if path and os.path.exists(path):
        for ln in open(path):
            m = re.match(r"^(v_\S+)\s+([0-9.]+)\s*$", ln)
            if m:
                costs[m.group(1)] = float(m.group(2))
    return costs


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="?")
    ap.add_argument("--kernel", default="raster_track_fused_kernel")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--costs", default=None)
    ap.add_argument("--min-valu", type=int, default=6, help="skip blocks with fewer VALU instructions")
    ap.add_argument("--list", action="store_true", help="print the instructions of every block shown")
    ap.add_argument("--segments", default=None, metavar="MNEMONIC",
                    help="instead of loop blocks: one row per SEGMENT of the instruction stream from one occurrence of MNEMONIC "
                         "(e.g. v_exp_f32: one per survivor body) to the next, all paths included")
    a = ap.parse_args()
    asm = a.asm or compile_asm()
    name, body = kernel_text(asm, a.kernel)
    over = load_costs(a.costs)
    print(f"# {name}")
    print(f"# blocks at loop depth >= {a.depth} with >= {a.min_valu} VALU instructions; cost = SIMD issue cycles per trip")
    total = collections.Counter()
    blist = list(blocks(body))
    if a.segments:
        flat = [(lab, mn, text) for lab, _, ins in blist for mn, text in ins]
        cuts = [i for i, (_, mn, _) in enumerate(flat) if mn.startswith(a.segments)]
        segs = []
        for k, c in enumerate(cuts):
            # a segment starts at the block boundary before the marker and ends before the next marker's block
            lo = c
            while lo > 0 and flat[lo - 1][0] == flat[c][0]:
                lo -= 1
            hi = cuts[k + 1] if k + 1 < len(cuts) else min(len(flat), c + 120)
            while k + 1 < len(cuts) and hi > lo and flat[hi - 1][0] == flat[cuts[k + 1]][0]:
                hi -= 1
            segs.append((flat[lo][0], a.depth, [(mn, text) for _, mn, text in flat[lo:hi]]))
        blist = segs
    for label, depth, ins in blist:
        if depth < a.depth:
            continue
        cnt = collections.Counter()
        cost = 0.0
        detail = collections.Counter()
        for mn, text in ins:
            c = classify(mn, text)
            cnt[c] += 1
            base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
            cost += over.get(base, CLASS_COST[c])
            if c in ("fma", "half", "pk", "trans", "mfma"):
                detail[base + (" (dpp)" if c == "half" and base.startswith(FULL_RATE) else "")] += 1
        n_valu = cnt["fma"] + cnt["half"] + cnt["pk"] + cnt["trans"] + cnt["mfma"]
        if n_valu < a.min_valu:
            continue
        total.update(cnt)
        cls = "  ".join(f"{k}={cnt[k]}" for k in ("fma", "half", "pk", "trans", "mfma", "lds", "vmem", "salu", "nop", "branch") if cnt[k])
        print(f"{label:<12} depth {depth}  VALU {n_valu:3d}  cost {cost:6.1f}   {cls}")
        print("             " + ", ".join(f"{k} x{v}" for k, v in sorted(detail.items(), key=lambda kv: -kv[1])))
        if a.list:
            for mn, text in ins:
                print(f"        {classify(mn, text):<6} {text}")
    print("# all listed blocks:", dict(total))


if __name__ == "__main__":
    main()
