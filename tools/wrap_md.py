#!/usr/bin/env python3
"""wrap_md.py FILE [WIDTH] -- re-flows a Markdown file to at most WIDTH (120) columns, in place.

Paragraphs and list items are re-wrapped (list items with a hanging indent); fenced code, headings, HTML and blank lines are left
alone.  A table whose rows fit stays a table; a table with a row wider than WIDTH becomes a list -- one item per row, the first cell
in bold, every further cell introduced by its column header -- because a Markdown table row cannot be broken.
Words are never split: a token longer than the width (a path, a long inline-code span) may still stick out.
"""
import re
import sys
import textwrap


def split_row(line):
    cells = [c.strip() for c in re.split(r"(?<!\\)\|", line.strip().strip("|"))]
    return cells


def wrap_item(prefix, text, width, indent):
    w = textwrap.TextWrapper(width=width, initial_indent=prefix, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)
    return w.wrap(text) or [prefix.rstrip()]


def flush_par(buf, out, width):
    if not buf:
        return
    first = buf[0]
    m = re.match(r"^(\s*)([*+-]|\d+[.)])\s+", first)
    if m:
        prefix = first[:m.end()]
        indent = " " * len(prefix)
        text = " ".join([first[m.end():].strip()] + [b.strip() for b in buf[1:]])
        out += wrap_item(prefix, text, width, indent)
    else:
        lead = re.match(r"^\s*", first).group(0)
        text = " ".join(b.strip() for b in buf)
        out += wrap_item(lead, text, width, lead)
    buf.clear()


def table_to_list(rows, out, width):
    header = split_row(rows[0])
    for r in rows[2:]:
        cells = split_row(r)
        parts = []
        for h, c in zip(header[1:], cells[1:]):
            if not c or c in ("–", "-", "—"):
                continue
            parts.append("%s: %s" % (h, c) if h else c)
        head = cells[0] if cells[0].startswith("**") or not cells[0] else "**%s**" % cells[0]
        text = head + (" — " + "; ".join(parts) if parts else "")
        out += wrap_item("* ", text, width, "  ")


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    lines = open(path).read().split("\n")
    out, buf, k = [], [], 0
    in_code = False
    while k < len(lines):
        ln = lines[k]
        if ln.strip().startswith("```"):
            flush_par(buf, out, width)
            in_code = not in_code
            out.append(ln)
            k += 1
            continue
        if in_code:
            out.append(ln)
            k += 1
            continue
        if ln.lstrip().startswith("|") and k + 1 < len(lines) and re.match(r"^\s*\|?\s*:?-{3,}", lines[k + 1]):
            flush_par(buf, out, width)
            rows = []
            while k < len(lines) and lines[k].lstrip().startswith("|"):
                rows.append(lines[k])
                k += 1
            if max(len(r) for r in rows) <= width:
                out += rows
            else:
                table_to_list(rows, out, width)
            continue
        if not ln.strip() or ln.startswith("#") or ln.lstrip().startswith("<"):
            flush_par(buf, out, width)
            out.append(ln)
            k += 1
            continue
        if re.match(r"^\s*([*+-]|\d+[.)])\s+", ln) and buf:
            flush_par(buf, out, width)  # a new list item ends the previous one
        if buf and re.match(r"^\s*", ln).group(0) != re.match(r"^\s*", buf[0]).group(0) and not re.match(r"^\s*([*+-]|\d+[.)])\s+", buf[0]):
            flush_par(buf, out, width)  # indentation changed inside plain text
        buf.append(ln)
        k += 1
    flush_par(buf, out, width)
    open(path, "w").write("\n".join(out))
    wide = [(i + 1, len(l)) for i, l in enumerate(out) if len(l) > width]
    print("%s: %d lines, %d wider than %d%s" % (path, len(out), len(wide), width, (" (first: line %d, %d columns)" % wide[0]) if wide else ""))


if __name__ == "__main__":
    main()
