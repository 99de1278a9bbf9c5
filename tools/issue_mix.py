"""python3 tools/issue_mix.py [LIBRARY] [KERNEL SUBSTRING ...]: the vector instructions of the library's kernels by issue
class, from the disassembly of its device code.

tools/micro/valu_issue.hip (profiles/r3_valu_issue.txt) measures two classes on gfx950 once two or more waves share a
SIMD: v_fma / v_fmac / v_add / v_sub / v_mul (f32), v_mov, v_and, v_add_u32 and v_cndmask reading vcc issue every 2.2
cycles; comparisons, min / max / min3 / max3 / med3, conversions, shifts, bit-field operations, v_cndmask reading an SGPR
pair, v_readlane, v_mul_lo, v_perm, packed multiply-adds issue every 4.1 cycles (v_rcp: 8).  Opcodes the microbenchmark
did not time are counted with the slow class and listed, so the share below errs towards "slow".

The count is static (every instruction of the kernel's text once).  For the search kernel the straight-line block of the
node step -- the one with the v_cvt_f32_ubyte decodes -- is also shown on its own: it is where two thirds of the
kernel's instructions are issued.  Default: pathtracing_amd/libpathtracing_amd.so, kernels search_kernel (the timed
instance: one mesh, no counting) and walk_bounce_kernel.  --json prints {kernel: {...}} for tools/summarize_sq.py."""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FAST_CYCLES, SLOW_CYCLES = 2.19, 4.09

FAST = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32"}  # the last two: v_add_u32's datapath; not timed themselves
SLOW_TIMED = {"v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3", "v_max3_f32", "v_min3_f32",
              "v_med3_f32", "v_min_f32", "v_max_f32", "v_lshlrev_b32", "v_mul_lo_u32", "v_bfe_u32", "v_cvt_f32_u32", "v_rcp_f32",
              "v_pk_fma_f32", "v_readlane_b32", "v_bfi_b32", "v_and_or_b32", "v_min_u32", "v_max_u32", "v_min3_u32", "v_perm_b32",
              "v_lshl_add_u32"}


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="issue_mix_")
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", copy], check=True, capture_output=True, cwd=tmp)
        code = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not code:
            raise SystemExit(f"{lib}: no device code object found")
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", os.path.join(tmp, code[0])], check=True,
                              capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, name = collections.OrderedDict(), None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            kernels[name] = []
            continue
        if name and line.startswith("\t"):
            ops = line.split("//")[0].split()
            if ops:
                kernels[name].append((ops[0], " ".join(ops[1:])))
    return kernels


def classify(op, args):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return "slow", True
    if base == "v_cndmask_b32":
        return ("fast", True) if op.endswith("_e32") or args.rstrip().endswith("vcc") else ("slow", True)
    if base in FAST:
        return "fast", True
    if base in SLOW_TIMED:
        return "slow", True
    return "slow", False  # not timed


def mix(insts):
    fast = slow = 0
    untimed = collections.Counter()
    for op, args in insts:
        if not op.startswith("v_"):
            continue
        cls, timed = classify(op, args)
        if cls == "fast":
            fast += 1
        else:
            slow += 1
        if not timed:
            untimed[re.sub(r"_(e32|e64)$", "", op)] += 1
    n = fast + slow
    return {"valu": n, "fast": fast, "slow": slow, "slow_class_share": slow / n if n else None,
            "mean_issue_cycles": (fast * FAST_CYCLES + slow * SLOW_CYCLES) / n if n else None,
            "untimed": dict(untimed.most_common()), "untimed_share": sum(untimed.values()) / n if n else None,
            "salu": sum(1 for op, _ in insts if op.startswith("s_")),
            "vmem": sum(1 for op, _ in insts if op.startswith(("global_", "flat_", "buffer_", "scratch_"))),
            "lds": sum(1 for op, _ in insts if op.startswith("ds_"))}


def node_step_block(insts):
    """The straight-line run of instructions around the byte-to-float decodes, cut at the branches on either side."""
    at = [i for i, (op, _) in enumerate(insts) if op.startswith("v_cvt_f32_ubyte")]
    if not at:
        return []
    lo, hi = at[0], at[-1]
    while lo > 0 and not insts[lo - 1][0].startswith(("s_cbranch", "s_branch")):
        lo -= 1
    while hi + 1 < len(insts) and not insts[hi][0].startswith(("s_cbranch", "s_branch")):
        hi += 1
    return insts[lo:hi + 1]


def main():
    args = [a for a in sys.argv[1:] if a != "--json"]
    as_json = "--json" in sys.argv[1:]
    lib = args[0] if args and os.path.exists(args[0]) else os.path.join(ROOT, "pathtracing_amd", "libpathtracing_amd.so")
    wanted = [a for a in args if a != lib] or ["search_kernelILb1ELb0E", "walk_bounce_kernelILb0E"]
    kernels = disassemble(lib)
    out = {}
    for name, insts in kernels.items():
        if not any(w in name for w in wanted):
            continue
        pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        rec = mix(insts)
        block = node_step_block(insts) if "search_kernel" in name else []
        if block:
            rec["node_step_block"] = mix(block)
        out[pretty] = rec
    if as_json:
        print(json.dumps(out))
        return
    print(f"{os.path.basename(lib)}\nissue classes: fast {FAST_CYCLES} cycles, slow {SLOW_CYCLES} cycles per wave64 instruction (profiles/r3_valu_issue.txt)")
    for pretty, rec in out.items():
        def show(tag, r):
            print(f"  {tag:24s} {r['valu']:5d} vector instructions: {r['fast']:5d} fast, {r['slow']:5d} slow "
                  f"({100 * r['slow_class_share']:.1f} % slow, mean {r['mean_issue_cycles']:.2f} cycles); "
                  f"{r['salu']} scalar, {r['vmem']} memory, {r['lds']} LDS; not timed (counted slow): "
                  f"{100 * r['untimed_share']:.1f} % {r['untimed']}")
        print(pretty)
        show("whole kernel (static)", rec)
        if "node_step_block" in rec:
            show("node step's block", rec["node_step_block"])


if __name__ == "__main__":
    main()
