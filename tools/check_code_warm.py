#!/usr/bin/env python3
"""Build-time bound for idc_warm_own_code (csrc/idc_kernels.h).

Every throughput / click kernel reads `lines` x 128 bytes of ITS OWN CODE as data at entry, starting at the 128-byte line of the
`s_getpc_b64` that idc_warm_own_code expands to.  The line counts are constants in the sources, chosen from one build's code
sizes; a compiler or flag change that emits smaller code could push the read past the end of the code object's .text.  This
script extracts the gfx950 code object of every built object file, finds each `s_getpc_b64` in the disassembly and checks

        (address of s_getpc & ~127) + lines(kernel instance) * 128  <=  end of .text

with `lines` evaluated from the call site's expression for the instance's template arguments (the demangled symbol).  Exit status 1 and a list of offenders otherwise.  Run by __graft_entry__.build() after the library
is built and by tests/test_round5_cpu.py.
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


def _top_level_split(text, sep=","):
    depth, parts, cur = 0, [], ""
    for ch in text:
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
            continue
        depth += ch in "(<"
        depth -= ch in ")>"
        cur += ch
    parts.append(cur)
    return parts


def _c_ternary_to_python(expr):
    """`a ? b : c ? d : e` -> python conditional expressions (right-associative, parentheses respected)"""
    expr = expr.strip()
    while expr.startswith("(") and expr.endswith(")"):              # (a ? b : c) -> a ? b : c when the parentheses wrap the whole
        d = 0
        for i, ch in enumerate(expr):
            d += ch == "("
            d -= ch == ")"
            if d == 0:
                break
        if i != len(expr) - 1:
            break
        expr = expr[1:-1].strip()
    depth = 0
    for i, ch in enumerate(expr):
        depth += ch == "("
        depth -= ch == ")"
        if ch == "?" and depth == 0:
            nest, d2 = 0, 0
            for j in range(i + 1, len(expr)):
                c = expr[j]
                d2 += c == "("
                d2 -= c == ")"
                if d2 == 0 and c == "?":
                    nest += 1
                elif d2 == 0 and c == ":":
                    if nest == 0:
                        return "((%s) if (%s) else (%s))" % (_c_ternary_to_python(expr[i + 1:j]), expr[:i], _c_ternary_to_python(expr[j + 1:]))
                    nest -= 1
    return expr


def call_sites(src):
    """[(kernel name, [template parameter names], python expression of `lines`)] for every idc_warm_own_code call of a source file"""
    text = open(src).read()
    kernels = [(m.start(), m.group(2), [p.split()[-1] for p in m.group(1).split(",")] if m.group(1) else [])
               for m in re.finditer(r"(?:template\s*<([^>]*)>\s*)?__global__\s+(?:void\s+)?(?:__launch_bounds__\s*\((?:[^()]|\([^()]*\))*\)\s*)?(?:void\s+)?(\w+)\s*\(", text)]
    # (round 6) kernel bodies shared by several kernels: `template <...> __device__ __forceinline__ void X_body(...)` called by thin
    # `__global__ void K(const ConvArgs a) { X_body<...>(a); }` wrappers -- a call site inside X_body belongs to every such K (same leading
    # template parameters)
    bodies = [(m.start(), m.group(2), [p.split()[-1] for p in m.group(1).split(",")])
              for m in re.finditer(r"template\s*<([^>]*)>\s*__device__\s+__forceinline__\s+void\s+(\w+_body)\s*\(", text)]
    wrappers = {}
    for m in re.finditer(r"void\s+(\w+)\s*\(const ConvArgs a\)\s*\{\s*(\w+_body)\s*<", text):
        wrappers.setdefault(m.group(2), []).append(m.group(1))
    out = []
    for m in re.finditer(r"idc_warm_own_code\(([^;]*)\);", text):
        if text.rfind("\n", 0, m.start()) >= 0 and "__device__" in text[text.rfind("\n", 0, m.start()):m.start()]:
            continue                                   # the definition itself
        parts = _top_level_split(m.group(1))
        if len(parts) < 3:
            continue
        owner = [k for k in kernels + bodies if k[0] < m.start()]
        if not owner:
            continue
        owner = max(owner, key=lambda k: k[0])
        for name in wrappers.get(owner[1], [owner[1]]):
            out.append((name, owner[2], _c_ternary_to_python(parts[2].strip())))
    return out


def lines_for(sites, demangled):
    """the `lines` value the kernel instance `demangled` (e.g. idc::conv_kwave_bf16<4, 2, 8>(...)) passes, or None"""
    for name, params, expr in sites:
        m = re.search(r"\b%s(?:<([^>]*)>)?\(" % re.escape(name), demangled)
        if not m:
            continue
        env = {}
        if m.group(1):
            for pname, val in zip(params, [v.strip() for v in m.group(1).split(",")]):
                env[pname] = {"true": 1, "false": 0}.get(val, None)
                if env[pname] is None:
                    env[pname] = int(re.sub(r"[^0-9-]", "", val) or 0)
        try:
            return int(eval(expr.replace("&&", " and ").replace("||", " or "), {"__builtins__": {}}, env))
        except Exception:
            return max(int(x) for x in re.findall(r"\d+", expr))         # cannot evaluate: the largest literal (conservative)
    return None


def check(obj, src):
    sites = call_sites(src)
    if not sites:
        return [], 0
    tmp = tempfile.mkdtemp(prefix="idc_cw_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cos:
            return ["%s: no gfx950 code object found" % obj], 0
        co = os.path.join(tmp, cos[0])
        sec = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", co], check=True, capture_output=True, text=True).stdout
        m = re.search(r"\.text\s+PROGBITS\s+([0-9a-f]+)\s+[0-9a-f]+\s+([0-9a-f]+)", sec)
        text_end = int(m.group(1), 16) + int(m.group(2), 16)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-C", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        bad, func, n = [], "?", 0
        for line in dis.splitlines():
            fm = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if fm:
                func = fm.group(1)
                continue
            if "s_getpc_b64" in line:
                am = re.search(r"//\s*([0-9A-Fa-f]+):", line) or re.match(r"^\s*([0-9a-f]+):", line)
                addr = int(am.group(1), 16)
                lines = lines_for(sites, func)
                if lines is None:
                    continue                               # an s_getpc of something else (none today)
                n += 1
                if (addr & ~127) + lines * 128 > text_end:
                    bad.append("%s: %s reads %d lines from 0x%x, .text ends at 0x%x (%d bytes short)"
                               % (os.path.basename(src), func.split("(")[0], lines, addr & ~127, text_end, (addr & ~127) + lines * 128 - text_end))
        if n == 0:
            bad.append("%s: calls idc_warm_own_code but no matching s_getpc_b64 was found in its code object" % os.path.basename(src))
        return bad, n
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    bad = []
    checked = 0
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip"):
            obj = os.path.join(CSRC, f[:-4] + ".o")
            if not os.path.exists(obj):
                bad.append("%s: not built" % obj)
                continue
            b, n = check(obj, os.path.join(CSRC, f))
            bad += b
            checked += n
    for b in bad:
        print("check_code_warm: " + b, file=sys.stderr)
    print("check_code_warm: %d kernel instances with own-code warm-up checked, %d problems" % (checked, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
