#!/usr/bin/env python3
"""Build-time hazard check for the hand-counted `s_waitcnt vmcnt` of idc_kw.hip.

conv_kwave_bf16 / conv_kwave_deconv_bf16 / conv_kwave_chain_bf16 issue their weight-fragment loads through `asm volatile` with a plain "=v" output
and wait for them later with counted `s_waitcnt vmcnt(N)` -- to the compiler the destination registers are defined at the asm statement, so a copy,
spill or re-allocation of one of them between the load and its wait (register pressure, a compiler upgrade) would silently read stale data.  This
script walks the disassembly of every kernel of the built object in program order with the in-order model the compiler itself uses for vmcnt on
gfx9-family targets: every VMEM instruction (global / buffer / scratch loads and stores, LDS-DMA) enters a FIFO; `s_waitcnt vmcnt(N)` retires the
oldest entries until N remain; a load's destination registers are PENDING until it retires.  Any other instruction that reads or writes a pending
register is reported, and so is any scratch (spill) traffic inside these kernels.  The walk is linear over the text (loops are seen once, in layout
order; the state is dropped at an unconditional branch, behind which the other arm of an if begins) -- the kernels' loads and waits pair up inside
straight-line tap sequences, which is what the check is for.

Run by __graft_entry__.build() and by tests/test_round5_cpu.py; exit status 1 and a list of findings otherwise.
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
VMEM = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, body):
    findings, fifo, pending = [], [], {}          # fifo of (is_load, dest regs); pending: reg -> line of its load
    n_asm_like = 0
    for ln, line in body:
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        operands = ins[len(op):]
        if op.startswith("scratch_"):
            findings.append("%s: line %d: %s (spill traffic in a kernel with hand-counted waits)" % (name, ln, ins[:60]))
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                while len(fifo) > keep:
                    _, dst = fifo.pop(0)
                    for r in dst:
                        pending.pop(r, None)
            continue
        if op == "s_endpgm":
            break
        if op == "s_branch":              # what follows is reached from elsewhere (the other arm of an if): the linear walk's state does not apply to it
            fifo, pending = [], {}
            continue
        if VMEM.match(op):
            to_lds = "load_lds" in op or re.search(r"\blds\b", operands) is not None          # LDS-DMA: no register destination
            is_load = "_load" in op and not to_lds
            parts = [p.strip() for p in operands.split(",")]
            dst = regs_of(parts[0]) if (is_load and parts) else set()
            srcs = regs_of(",".join(parts[1:] if is_load else parts))
            bad = srcs & set(pending)
            if bad:
                findings.append("%s: line %d: %s uses v%s while the load of line %d is outstanding" % (name, ln, ins[:70], sorted(bad)[:4], pending[sorted(bad)[0]]))
            # a load may target registers that are still pending (WAW on the in-order return path is safe), but then the older entry no longer owns them
            for r in dst:
                pending[r] = ln
            fifo.append((is_load, dst))
            n_asm_like += is_load
            continue
        used = regs_of(operands) & set(pending)
        if used:
            findings.append("%s: line %d: %s touches v%s before the load of line %d was waited for" % (name, ln, ins[:70], sorted(used)[:4], pending[sorted(used)[0]]))
    return findings, n_asm_like


def main():
    obj = os.path.join(CSRC, "idc_kw.o")
    if not os.path.exists(obj):
        print("check_vmem_hazards: %s not built" % obj, file=sys.stderr)
        return 1
    tmp = tempfile.mkdtemp(prefix="idc_vh_")
    try:
        shutil.copy(obj, tmp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.join(tmp, "idc_kw.o")], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            print("check_vmem_hazards: no gfx950 code object", file=sys.stderr)
            return 1
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-C", "--no-show-raw-insn", os.path.join(tmp, co[0])],
                             check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, cur = {}, None
    for ln, line in enumerate(dis.splitlines(), 1):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1).split("(")[0]
            kernels[cur] = []
            continue
        if cur is not None and line.startswith("\t"):
            kernels[cur].append((ln, line))
    findings, checked, loads = [], 0, 0
    for name, body in kernels.items():
        if "conv_kwave" not in name:
            continue
        f, n = check_kernel(name, body)
        findings += f
        checked += 1
        loads += n
    for f in findings:
        print("check_vmem_hazards: " + f, file=sys.stderr)
    print("check_vmem_hazards: %d kernels, %d register loads walked, %d findings" % (checked, loads, len(findings)))
    return 1 if findings or checked < 8 else 0


if __name__ == "__main__":
    sys.exit(main())
