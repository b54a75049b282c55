"""Test infrastructure (not product code): records the public surface of the reference's model wrapper that callers touch,
by parsing (ast, never importing: cv2 / skimage / caffe are absent) /root/reference/data/colorize_image.py, ideepcolor.py and
the notebooks -> tests/golden/reference_ci_surface.json.  tests/test_round4_cpu.py asserts that every name resolves through
both import routes of this package with the reference's constructor / prep_net signatures."""
import ast
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_ci_surface.json")


def signature(fn):
    a = fn.args
    names = [x.arg for x in a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [ast.literal_eval(d) if not isinstance(d, ast.UnaryOp) else ast.literal_eval(d) for d in a.defaults]
    return [[n, (repr(d) if i >= len(names) - len(a.defaults) else None)] for i, (n, d) in enumerate(zip(names, defaults))]


def main():
    src = open(os.path.join(REF, "data", "colorize_image.py")).read()
    tree = ast.parse(src)
    classes, functions = {}, []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            functions.append(node.name)
        if isinstance(node, ast.ClassDef):
            methods = {}
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and not m.name.startswith("__patch") and not m.name.startswith("_ColorizeImage"):
                    methods[m.name] = signature(m)
            classes[node.name] = {"bases": [b.id for b in node.bases if isinstance(b, ast.Name)], "methods": methods}
    used = set()
    callers = [os.path.join(REF, "ideepcolor.py")] + [os.path.join(REF, f) for f in sorted(os.listdir(REF)) if f.endswith(".ipynb")]
    for path in callers:
        text = open(path).read()
        used.update(re.findall(r"\bCI\.([A-Za-z_][A-Za-z0-9_]*)", text))
    json.dump({"source": "data/colorize_image.py, ideepcolor.py, *.ipynb of junyanz/interactive-deep-colorization (parsed, not imported)",
               "functions": functions, "classes": classes, "names_used_by_callers": sorted(used)},
              open(OUT, "w"), sort_keys=True)
    print("wrote", OUT, sorted(used))


if __name__ == "__main__":
    main()
