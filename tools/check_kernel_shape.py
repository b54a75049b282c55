#!/usr/bin/env python3
"""Build-time shape check of the throughput kernels' code (round 5).

The store epilogues of conv_igemm_v2p / conv_igemm_v2m / conv_ds_fused_m / conv1_block_fused_t once compiled to a uniform branch per packed pair
(`if (has_bn)` / `if (a.act == 1)` inside fully unrolled element loops are not unswitched), every transposed LDS read sunk under its store's bounds check and
scratch traffic for a small array -- 764 branches and 8.5 k lines in one kernel, 2.6 k cycles per stored pixel row, 2 % of the N = 32 forward
(profiles/r05_epilogue.txt).  Nothing in a test notices that: the results are the same.  This script disassembles the built objects and holds every shipped
throughput kernel to a budget of conditional branches and scratch instructions taken from the build that fixed it (+ 50 %), so that the same thing coming back
-- after an edit or a compiler upgrade -- fails the build instead of costing a few percent silently.

Run by __graft_entry__.build() and by tests/test_round5_cpu.py.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "interactive_deep_colorization_amd", "csrc")
def _llvm_bin():
    """llvm-objdump / llvm-readelf of the toolchain that built the objects: $IDC_LLVM_BIN, else beside $HIPCC, else under $ROCM_PATH,
    else /opt/rocm (no hard-coded install layout)."""
    import shutil as _sh
    cands = [os.environ.get("IDC_LLVM_BIN")]
    hipcc = os.environ.get("HIPCC") or _sh.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    for var in ("ROCM_PATH", "ROCM_HOME"):
        if os.environ.get(var):
            cands.append(os.path.join(os.environ[var], "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if c and os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    return cands[-1]


LLVM = _llvm_bin()
# kernel name prefix -> (object, max conditional branches, max scratch instructions); measured at the round-5 HEAD: v2p 92-109, v2m 119-198, ds 63-66, conv1 block 36-47; scratch 0 (v2p<4,2,*>: 6 = three spilled epilogue constants)
BUDGET = [("conv_igemm_v2p<2, 2,", "idc_v2m.o", 165, 0), ("conv_igemm_v2p<4, 2,", "idc_v2m.o", 140, 12), ("conv_igemm_v2m<", "idc_v2m.o", 300, 0),
          ("conv_ds_fused_m<", "idc_dsm.o", 100, 0), ("conv1_block_fused_t<4,", "idc_conv1.o", 70, 0),
          # round 6, the operand-split forms: their K loops are spill-free (the scratch instructions are split_epilogue's fp32 constants: 49-56 measured);
          # conv_ds_fused_ms once had 21 of them INSIDE its K loops (the S halo's 44 prefetch registers) -- that build counted 102
          ("conv_ds_fused_ms", "idc_dsm.o", 460, 75), ("conv1_2_split_kernel<", "idc_conv1.o", 65, 0), ("conv1_1_split_kernel<", "idc_conv1.o", 420, 85)]


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="idc_ks_")
    try:
        shutil.copy(obj, tmp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.join(tmp, os.path.basename(obj))], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-C", "--no-show-raw-insn", os.path.join(tmp, co[0])],
                              check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_stats(dis):
    """{kernel name: (conditional branches, scratch instructions, instructions)} of one disassembly"""
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = [0, 0, 0]
            continue
        if cur is None or not line.startswith("\t"):
            continue
        op = line.split()[0] if line.split() else ""
        out[cur][2] += 1
        if op.startswith("s_cbranch"):
            out[cur][0] += 1
        elif op.startswith("scratch_"):
            out[cur][1] += 1
    return out


def check(stats_by_obj):
    findings, seen = [], 0
    for prefix, obj, max_br, max_scr in BUDGET:
        hits = [(k, v) for k, v in stats_by_obj.get(obj, {}).items() if ("idc::" + prefix) in k or k.startswith(prefix) or (" " + prefix) in k]
        if not hits:
            findings.append("no kernel matching %r in %s" % (prefix, obj))
        for k, (br, scr, n) in hits:
            seen += 1
            if br > max_br:
                findings.append("%s: %d conditional branches in %d instructions (budget %d): a run-time `if` inside an unrolled element loop?" % (k, br, n, max_br))
            if scr > max_scr:
                findings.append("%s: %d scratch instructions (budget %d): an array that did not stay in registers?" % (k, scr, max_scr))
    return findings, seen


def main():
    stats = {}
    for obj in sorted(set(b[1] for b in BUDGET)):
        path = os.path.join(CSRC, obj)
        if not os.path.exists(path):
            print("check_kernel_shape: %s not built" % path, file=sys.stderr)
            return 1
        stats[obj] = kernel_stats(disassemble(path))
    findings, seen = check(stats)
    for f in findings:
        print("check_kernel_shape: " + f, file=sys.stderr)
    print("check_kernel_shape: %d kernels held to their branch / scratch budgets, %d findings" % (seen, len(findings)))
    if (findings or seen < 8) and os.environ.get("IDC_SHAPE_CHECK", "").lower() in ("warn", "0", "off"):
        # a performance lint, not a correctness check: a new compiler may legitimately move the numbers -- the build goes on, loudly
        print("check_kernel_shape: IDC_SHAPE_CHECK=%s -- findings reported above, NOT failing the build" % os.environ["IDC_SHAPE_CHECK"], file=sys.stderr)
        return 0
    return 1 if findings or seen < 8 else 0


if __name__ == "__main__":
    sys.exit(main())
