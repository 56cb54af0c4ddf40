#!/usr/bin/env python3
"""Reflow the plain paragraphs and list items of a markdown file at 130 characters: tools/wrap_md.py FILE...
Tables, headings, fenced and indented code, and lines that are already short enough in a block with nothing over-long are left as they are."""
import re
import sys
import textwrap

WIDTH = 130
ITEM = re.compile(r'^(\s*)([*+-]|\d+[.)])\s+')


def wrap_block(lines):
    if not any(len(l) > WIDTH for l in lines):
        return lines
    out, cur, indent, first = [], [], '', ''

    def flush():
        if cur:
            out.extend(textwrap.fill(' '.join(cur), WIDTH, initial_indent=first, subsequent_indent=indent, break_long_words=False,
                                     break_on_hyphens=False).split('\n'))
    for l in lines:
        m = ITEM.match(l)
        if m:
            flush()
            first, indent, cur = m.group(0), ' ' * len(m.group(0)), [l[m.end():].strip()]
        elif not cur:
            lead = re.match(r'^\s*', l).group(0)
            first, indent, cur = lead, lead, [l.strip()]
        else:
            cur.append(l.strip())
    flush()
    return out


def wrap(text):
    out, block, fenced = [], [], False
    for l in text.split('\n'):
        fence = l.lstrip().startswith('```')
        plain = not fenced and not fence and l.strip() and not l.lstrip().startswith(('|', '#', '>')) and not (l.startswith('    ') and not block)
        if plain:
            block.append(l)
            continue
        out.extend(wrap_block(block)); block = []
        out.append(l)
        if fence:
            fenced = not fenced
    out.extend(wrap_block(block))
    return '\n'.join(out)


for path in sys.argv[1:]:
    s = open(path).read()
    t = wrap(s)
    if t != s:
        open(path, 'w').write(t)
    print(path, sum(1 for l in t.split('\n') if len(l) > WIDTH and not l.lstrip().startswith('|')), 'plain lines still over', WIDTH)
