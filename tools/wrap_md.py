"""Re-wraps the prose of a markdown file to a column limit (tables, headings, fenced code and blank lines stay as they are; a
bullet's continuation lines keep its indentation).  usage: python tools/wrap_md.py FILE [width=118]"""
import re, sys, textwrap
path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
out, para, fence = [], [], False
def flush():
    if not para: return
    first = para[0]
    m = re.match(r"^(\s*)([*-]|\d+\.)\s+", first)
    ind = m.group(0) if m else re.match(r"^\s*", first).group(0)
    sub = " " * len(ind)
    text = " ".join([first[len(ind):].strip()] + [l.strip() for l in para[1:]])
    out.extend(textwrap.wrap(text, width=width, initial_indent=ind, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False))
    para.clear()
for line in open(path).read().split("\n"):
    if line.startswith("```"):
        flush(); fence = not fence; out.append(line); continue
    if fence or not line.strip() or line.startswith("#") or line.lstrip().startswith("|") or line.startswith("---"):
        flush(); out.append(line); continue
    if re.match(r"^\s*([*-]|\d+\.)\s+", line) or (para and re.match(r"^\S", line) and re.match(r"^\s", para[0]) and not re.match(r"^\s*([*-]|\d+\.)\s+", para[0])):
        flush()
    para.append(line)
flush()
open(path, "w").write("\n".join(out))
