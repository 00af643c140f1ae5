#!/usr/bin/env python3
"""Price a kernel's vector instructions with the issue costs tools/micro/valu_ops.hip measured on gfx950.

gfx950 issues a wave64 vector instruction in one quad-cycle; two instructions of the simple class (v_mov / and / or /
xor / add / sub / lshr / ashr / bitop3 with VGPR, inline-constant or literal operands) from two waves of a SIMD can share
a quad-cycle (counter SQ_ACTIVE_INST_VALU2), everything else - left shifts, compares, selects, min/max, 24-bit
multiplies, bit counts, every three-operand form, anything with an SGPR operand - takes the quad-cycle alone
(profiles/r02_valu_ops.txt).  This tool lists a kernel's basic blocks with their share of pairable instructions.

  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ivapor_amd/csrc --cuda-device-only -S -o /tmp/vapor.s vapor_amd/csrc/vapor_hip.hip
  python tools/isa_cost.py /tmp/vapor.s join_kernelINS_7JoinBigELi2ELi10 [--blocks] [--ops]
"""
import collections
import re
import sys

PAIRABLE = {"v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
            "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32"}
SREG = re.compile(r"\b(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0|scc)\b")


def base_op(op: str) -> str:
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if op.endswith(suf):
            return op[: -len(suf)]
    return op


def pairable(op: str, operands: str) -> bool:
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return False
    return base_op(op) in PAIRABLE and not SREG.search(operands)


def kernel_body(path: str, needle: str):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if needle in l and l.rstrip().endswith(":") or (needle in l and ": ; @" in l))
    out = []
    for l in lines[start + 1:]:
        if l.startswith("\t.amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        out.append(l)
    return out


def main():
    path, needle = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    show_ops = "--ops" in sys.argv
    body = kernel_body(path, needle)
    blocks = []
    cur = {"label": "entry", "ins": []}
    for l in body:
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            if s.startswith(".LBB") and s.endswith(":"):
                blocks.append(cur)
                cur = {"label": s[:-1], "ins": []}
            continue
        if s.endswith(":"):
            blocks.append(cur)
            cur = {"label": s[:-1], "ins": []}
            continue
        parts = s.split(None, 1)
        cur["ins"].append((parts[0], parts[1] if len(parts) > 1 else ""))
    blocks.append(cur)
    tot = collections.Counter()
    ops = collections.Counter()
    for b in blocks:
        c = collections.Counter()
        for op, rest in b["ins"]:
            if op.startswith("v_"):
                k = "valu_pair" if pairable(op, rest) else "valu_solo"
                ops[(base_op(op), k)] += 1
            elif op.startswith("ds_"):
                k = "lds"
            elif op.startswith("s_waitcnt"):
                k = "wait"
            elif op.startswith("s_cbranch") or op.startswith("s_branch"):
                k = "branch"
            elif op.startswith("s_"):
                k = "salu"
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
                k = "vmem"
            else:
                k = "other"
            c[k] += 1
        b["c"] = c
        tot.update(c)
        if show_blocks and sum(c.values()) >= 6:
            tgt = [r for o, r in b["ins"] if o.startswith("s_cbranch") or o.startswith("s_branch")]
            print(f"{b['label']:<12} valu {c['valu_pair'] + c['valu_solo']:4d} (pairable {c['valu_pair']:4d})  lds {c['lds']:3d}  salu {c['salu']:3d}  "
                  f"vmem {c['vmem']:2d}  wait {c['wait']:2d}  -> {','.join(tgt)}")
    v = tot["valu_pair"] + tot["valu_solo"]
    print(f"total: valu {v} (pairable {tot['valu_pair']} = {tot['valu_pair'] / max(v, 1):.2f}), lds {tot['lds']}, salu {tot['salu']}, vmem {tot['vmem']}, waits {tot['wait']}")
    if show_ops:
        for (op, k), n in sorted(ops.items(), key=lambda kv: -kv[1])[:60]:
            print(f"  {n:5d}  {op:<24} {k}")


if __name__ == "__main__":
    main()
