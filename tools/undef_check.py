"""Poor man's pyflakes (no linters in the image): names loaded in a function that are neither assigned in it or an
enclosing function, nor module globals, nor builtins.   python tools/undef_check.py FILE..."""
import ast
import builtins
import sys


def names_assigned(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            out.add(n.name)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
    return out


def check(path):
    tree = ast.parse(open(path).read())
    glob = set(dir(builtins)) | {"__file__", "__name__"}
    for top in tree.body:  # module level only: what a function can really see
        if isinstance(top, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            glob.add(top.name)
        else:
            glob |= names_assigned(top)
    bad = 0

    def visit(fn, outer):
        nonlocal bad
        local = names_assigned(fn) | outer
        nested = [n for n in ast.walk(fn) if n is not fn and isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda))]
        inside_nested = set()
        for nf in nested:
            for n in ast.walk(nf):
                if n is not nf:
                    inside_nested.add(id(n))
        for n in ast.walk(fn):
            if id(n) in inside_nested:
                continue
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in local and n.id not in glob:
                print("%s:%d: undefined name %r in %s" % (path, n.lineno, n.id, getattr(fn, "name", "<lambda>")))
                bad += 1
        for nf in nested:
            if id(nf) not in inside_nested:  # direct children only; deeper ones are reached recursively
                visit(nf, local)

    for top in tree.body:
        if isinstance(top, (ast.FunctionDef, ast.AsyncFunctionDef)):
            visit(top, set())
        elif isinstance(top, ast.ClassDef):
            for m in top.body:
                if isinstance(m, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    visit(m, set())
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
