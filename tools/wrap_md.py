#!/usr/bin/env python3
"""Re-flows a Markdown file's prose to at most N columns (default 120) without touching tables, fenced code, headings, HTML comments
or the generated blocks: paragraphs and list items are re-wrapped with their own indentation, nothing else moves.

    python tools/wrap_md.py DESIGN.md [120] [--check]      # --check: exit 1 and list the lines that are too long
"""
import re
import sys
import textwrap

BULLET = re.compile(r"^(\s*)([-*+]|\d+[.)])\s+")


def untouchable(line):
    s = line.lstrip()
    return (not s) or s.startswith(("|", "#", "```", "<!--", ">")) or s.startswith("    ") and not BULLET.match(line)


def wrap(text, width):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = BULLET.match(first)
        indent = re.match(r"^\s*", first).group(0)
        if m:
            head = m.group(0)
            body = first[len(head):]
            rest = " " * len(head)
        else:
            head, body, rest = indent, first[len(indent):], indent
        words = " ".join([body] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(words, width=width, initial_indent=head, subsequent_indent=rest, break_long_words=False,
                                 break_on_hyphens=False))
        para.clear()
    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence:
            out.append(line); continue
        if untouchable(line):
            flush(); out.append(line); continue
        if BULLET.match(line):
            flush(); para.append(line); continue
        if para:
            # a continuation line belongs to the paragraph if it is indented like its body (or the paragraph is plain prose)
            para.append(line)
        else:
            para.append(line)
    flush()
    return "\n".join(out)


def too_long(text, width):
    bad, fence = [], False
    for i, line in enumerate(text.split("\n"), 1):
        if line.lstrip().startswith("```"):
            fence = not fence
            continue
        if fence or line.lstrip().startswith(("|", "<!--")) or "](" in line and len(line.split()) == 1:
            continue
        if len(line) > width:
            bad.append((i, len(line)))
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path, width = args[0], int(args[1]) if len(args) > 1 else 120
    text = open(path).read()
    if "--check" in sys.argv:
        bad = too_long(text, width)
        for i, n in bad:
            print(f"{path}:{i}: {n} columns")
        sys.exit(1 if bad else 0)
    open(path, "w").write(wrap(text, width))


if __name__ == "__main__":
    main()
