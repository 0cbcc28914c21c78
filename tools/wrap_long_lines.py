"""Wraps source lines longer than LIMIT characters: breaks behind a comma at bracket depth >= 1, or splits a plain string literal at a
space into two adjacent literals (implicit concatenation inside brackets).  Checks that the AST is unchanged.
usage: python tools/wrap_long_lines.py [--limit 140] file.py ..."""
import ast
import io
import sys
import tokenize

LIMIT = 140
files = sys.argv[1:]
if files and files[0] == '--limit':
    LIMIT = int(files[1])
    files = files[2:]


def wrap_once(src):
    lines = src.split('\n')
    toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
    depth_at = {}
    depth = 0
    for t in toks:
        if t.type == tokenize.OP and t.string in '([{':
            depth += 1
        elif t.type == tokenize.OP and t.string in ')]}':
            depth -= 1
        depth_at[(t.start, t.end)] = depth
    for ln, line in enumerate(lines, 1):
        if len(line) <= LIMIT:
            continue
        on = [t for t in toks if t.start[0] == ln and t.end[0] == ln]
        indent = len(line) - len(line.lstrip())
        cont = ' ' * (indent + 4)
        # 1. last comma (depth >= 1) that ends before the limit and leaves something behind it
        best = None
        for t in on:
            if t.type == tokenize.OP and t.string == ',' and depth_at[(t.start, t.end)] >= 1 and t.end[1] <= LIMIT - 2 and line[t.end[1]:].strip() \
                    and not line[t.end[1]:].lstrip().startswith('#'):
                best = t
        if best is not None and best.end[1] > indent + 20:
            lines[ln - 1] = line[:best.end[1]].rstrip()
            lines.insert(ln, cont + line[best.end[1]:].lstrip())
            return '\n'.join(lines), True
        # 2. a trailing comment: move it onto its own line in front
        for t in on:
            if t.type == tokenize.COMMENT and t.start[1] > indent:
                lines[ln - 1] = line[:t.start[1]].rstrip()
                lines.insert(ln - 1, ' ' * indent + t.string)
                return '\n'.join(lines), True
        # 3. split a plain string literal that crosses the limit (inside brackets)
        for t in on:
            if t.type == tokenize.STRING and t.start[1] < LIMIT - 20 < t.end[1] and depth_at[(t.start, t.end)] >= 1:
                s = t.string
                q = s[-1]
                prefix = s[:s.index(q)]
                if s.endswith(q * 3) or 'f' in prefix.lower() or 'r' in prefix.lower():
                    continue
                cut = line.rfind(' ', t.start[1] + len(prefix) + 2, LIMIT - 2)
                if cut <= t.start[1] + len(prefix) + 1 or line[cut - 1] == '\\':
                    continue
                lines[ln - 1] = line[:cut + 1] + q
                lines.insert(ln, ' ' * t.start[1] + prefix + q + line[cut + 1:])
                return '\n'.join(lines), True
        # 4. a comment-only line: wrap at a space
        if line.lstrip().startswith('#'):
            cut = line.rfind(' ', indent + 10, LIMIT)
            if cut > 0:
                lines[ln - 1] = line[:cut].rstrip()
                lines.insert(ln, ' ' * indent + '# ' + line[cut + 1:])
                return '\n'.join(lines), True
        print('  cannot wrap line %d (%d chars)' % (ln, len(line)))
    return '\n'.join(lines), False


for path in files:
    src = open(path).read()
    before = ast.dump(ast.parse(src))
    n = 0
    while True:
        new, changed = wrap_once(src)
        if not changed:
            break
        src, n = new, n + 1
        if n > 2000:
            break
    assert ast.dump(ast.parse(src)) == before, 'AST changed in %s' % path
    open(path, 'w').write(src)
    print(path, 'wrapped %d times; longest line now %d' % (n, max(len(x) for x in src.split('\n'))))
