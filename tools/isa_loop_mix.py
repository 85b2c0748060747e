"""Static instruction mix of a kernel's loops, from the shipped code object (no GPU needed).

    python tools/isa_loop_mix.py <object basename, e.g. attention_mfma> <kernel name substring> [--min-instr N] [--dump]

Disassembles the gfx950 code object inside pytorch-generative_amd/build/<name>.o (`llvm-objdump --offloading`, `-d`),
finds every loop of the kernel (a backward branch: body = target .. branch) and prints, per loop, the number of
instructions by issue class together with an ISSUE-CYCLE estimate from the per-instruction costs measured on MI355X
(tools/exp/ubench.hip, tools/exp/coexec_ubench.hip; profiles/README.md rounds 1 and 4):

    v_mfma_f32_16x16x32_bf16 16 (co-executes with other waves' VALU)   v_mfma_f32_16x16x4_f32 33   v_mfma_f32_4x4x1 10.5
    v_exp / v_log / v_rcp / v_rsq / v_sqrt (quarter rate) 16 issue, measured 10-12 back to back       plain VALU 4 (wave64)
    v_pk_* fp32 4 (two results per lane)     ds_read/ds_write b128 8-13, smaller 4      s_* 1 (own issue port)

The estimate is an upper bound on what ONE wave needs from its SIMD's issue ports if nothing overlapped; it is meant for
comparing two versions of a loop and for reading off which class dominates, not as a timing model.
"""

import argparse
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.environ.get("PG_OBJ_DIR") or os.path.join(ROOT, "pytorch-generative_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        if "16x16x32" in op or "32x32x16" in op:
            return "mfma_bf16"
        if "4x4x1" in op:
            return "mfma_4x4"
        return "mfma_f32"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "valu_cmp"
    if op.startswith(("ds_read", "ds_load")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store")):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_store"
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_atomic"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


COST = {"mfma_bf16": 16, "mfma_f32": 33, "mfma_4x4": 10.5, "trans": 11, "valu": 4, "valu_pk": 4, "valu_cmp": 4,
        "lds_read": 8, "lds_write": 8, "lds_other": 4, "vmem_load": 4, "vmem_store": 4, "vmem_atomic": 4}


def disassemble(obj_base):
    obj = os.path.join(OBJ, obj_base + ".o")
    if not os.path.exists(obj):
        sys.exit(f"{obj} not found (python pytorch-generative_amd/build.py first)")
    for old in glob.glob(obj + ".0.*"):
        os.remove(old)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", obj], check=True, capture_output=True)
    co = glob.glob(obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950")
    if not co:
        sys.exit("no gfx950 code object in " + obj)
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "-C", co[0]], check=True,
                          capture_output=True, text=True).stdout
    for f in glob.glob(obj + ".0.*"):
        os.remove(f)
    return text


def kernels(text):
    """{demangled name: [(addr, opcode, operands, branch target address or None)]}"""
    out, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            base = int(m.group(1), 16)
            cur = out.setdefault(m.group(2), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<.*\+0x([0-9a-f]+)>)?\s*$", line)
        if m and cur is not None:
            tgt = base + int(m.group(4), 16) if m.group(4) else None
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), tgt))
    return out


def loops(instrs):
    """[(start index, end index)] for every backward branch"""
    index_of = {addr: i for i, (addr, _, _, _) in enumerate(instrs)}
    found = []
    for i, (addr, op, _, tgt) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")) and tgt is not None and tgt <= addr and tgt in index_of:
            found.append((index_of[tgt], i))
    return found


def mix(instrs, a, b):
    counts = {}
    for _, op, _, _ in instrs[a:b + 1]:
        c = classify(op)
        counts[c] = counts.get(c, 0) + 1
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("kernel")
    ap.add_argument("--min-instr", type=int, default=40)
    ap.add_argument("--dump", action="store_true", help="print the body of the largest loop (or of the loop --at)")
    ap.add_argument("--at", type=lambda x: int(x, 0), default=None, help="start address of the loop to report (with --dump: to print)")
    args = ap.parse_args()
    ks = kernels(disassemble(args.obj))
    names = [n for n in ks if args.kernel in n]
    if not names:
        sys.exit("no kernel matches; have:\n  " + "\n  ".join(sorted(ks)))
    for n in names:
        ins = ks[n]
        print(f"== {n}: {len(ins)} instructions")
        ls = [(a, b) for a, b in loops(ins) if b - a + 1 >= args.min_instr]
        # innermost first: a loop that contains another is reported after it
        for a, b in sorted(ls, key=lambda ab: ab[1] - ab[0]):
            if args.at is not None and not args.dump and ins[a][0] != args.at:
                continue
            c = mix(ins, a, b)
            est = sum(COST.get(k, 0) * v for k, v in c.items())
            coexec = COST["mfma_bf16"] * c.get("mfma_bf16", 0)
            inner = [(x, y) for x, y in ls if (x, y) != (a, b) and a <= x and y <= b]
            print(f"  loop @{ins[a][0]:#x}..{ins[b][0]:#x}: {b - a + 1} instr"
                  + (f" (contains {len(inner)} inner loop(s))" if inner else ""))
            print("    " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -COST.get(kv[0], 0) * kv[1])))
            print(f"    issue estimate {est:.0f} cycles, of which {coexec:.0f} on the bf16 matrix pipe (co-executing)")
        if args.dump and ls:
            a, b = max(ls, key=lambda ab: ab[1] - ab[0])
            if args.at is not None:
                a, b = min((ab for ab in ls if ins[ab[0]][0] == args.at), key=lambda ab: ab[1] - ab[0])
            for addr, op, a_, tgt in ins[a:b + 1]:
                print(f"      {addr:#08x} {op} {a_}" + (f"   -> {tgt:#x}" if tgt else ""))


if __name__ == "__main__":
    main()
