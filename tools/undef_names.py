"""Poor man's pyflakes (none in the image): names a module loads but never binds anywhere (imports, defs, assignments, arguments,
comprehension / with / except / for targets).  usage: python tools/undef_names.py file.py ..."""
import ast
import builtins
import sys

for path in sys.argv[1:]:
    tree = ast.parse(open(path).read())
    bound = set(dir(builtins))
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            bound.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                bound.add(x.arg)
        if isinstance(n, ast.Import):
            for x in n.names:
                bound.add((x.asname or x.name).split('.')[0])
        if isinstance(n, ast.ImportFrom):
            for x in n.names:
                bound.add(x.asname or x.name)
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        if isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        if isinstance(n, (ast.Global, ast.Nonlocal)):
            bound.update(n.names)
    missing = sorted({(n.id, n.lineno) for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound})
    print(path, 'OK' if not missing else 'UNDEFINED: %s' % missing)
