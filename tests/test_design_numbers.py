"""Evidence hygiene: every "x.xxx ms" that DESIGN.md attributes to a committed
profile must be IN that profile.

Attribution rule (the convention DESIGN.md is written to): a sentence that cites one or more
`profiles/...` files attributes every millisecond figure OF THAT SENTENCE to them -- each must be
found in at least one of the cited files.  Sentences without a citation are prose, not evidence,
and are not checked.  A figure is found when some number in the file equals it after rounding to
the figure's own decimals; kernel-stats CSVs hold nanoseconds and some tables microseconds, so a
file number also counts divided by 1e3 or 1e6.  Figures of three or more decimals only (coarser
ones are summaries)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r"profiles/[A-Za-z0-9_./-]*[A-Za-z0-9]\.(?:txt|csv|json|tsv|md)")
# every figure of three or four decimals in a citing sentence that speaks of milliseconds -- not only the one the
# unit follows: "`T` 0.7831, `X` 0.7889 and `P` 0.7846 ms" attributes three figures.  Percentages, exponents
# (2.9e-7) and parts of longer numbers are left out.
MS = re.compile(r"(?<![\d.\w-])(\d+\.\d{3,4})(?![\d%]|\s*%|e[-+]?\d)")
NUM = re.compile(r"(?<![A-Za-z_])[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?")


def sentences(text):
    """Bullets and paragraphs first, then sentence ends: '. ' followed by a capital, a backtick or a bracket."""
    out = []
    for block in re.split(r"\n\s*\n|\n(?=\s*[-*] )|\n(?=\|)", text):
        block = " ".join(block.split())
        out += re.split(r"(?<=[.;:])\s+(?=[A-Z`(\[*])", block)
    return [s for s in out if s]


def file_numbers(path):
    vals = set()
    for tok in NUM.findall(open(path, errors="replace").read()):
        try:
            v = float(tok)
        except ValueError:
            continue
        vals.add(v)
    return vals


def found(q, decimals, vals):
    for v in vals:
        for scale in (1.0, 1e-3, 1e-6):
            if abs(round(v * scale, decimals) - q) < 0.5 * 10 ** -decimals * 1e-6 + 1e-12:
                return True
    return False


DOCS = ("DESIGN.md", "profiles/r04/NOTES.md")      # (round 5: the allocation-mode narrative and round 3's closed leads moved to NOTES.md)


def claims():
    text = "\n\n".join(open(os.path.join(ROOT, d)).read() for d in DOCS)
    for s in sentences(text):
        cites = CITE.findall(s)
        if not cites:
            continue
        if not re.search(r"\bms\b", s):
            continue
        for m in MS.finditer(s):
            yield s, cites, m.group(1)


def test_every_ms_figure_attributed_to_a_profile_is_in_that_profile():
    cache, bad, checked = {}, [], 0
    for s, cites, fig in claims():
        files = [os.path.join(ROOT, c) for c in cites]
        missing = [f for f in files if not os.path.exists(f)]
        if missing:
            bad.append(("cited file does not exist", os.path.relpath(missing[0], ROOT), s[:160]))
            continue
        vals = set()
        for f in files:
            if f not in cache:
                cache[f] = file_numbers(f)
            vals |= cache[f]
        checked += 1
        if not found(float(fig), len(fig.split(".")[1]), vals):
            bad.append((fig + " ms not in", ", ".join(cites), s[:200]))
    assert checked >= 20, "DESIGN.md attributes too few figures to profiles: the convention is not being used"
    assert not bad, "\n".join("  %s %s  <<%s>>" % b for b in bad)


def test_cited_profiles_exist():
    text = "\n\n".join(open(os.path.join(ROOT, d)).read() for d in DOCS)
    missing = sorted({c for c in CITE.findall(text) if not os.path.exists(os.path.join(ROOT, c))})
    assert not missing, missing


SHAPE_HDR = re.compile(r"^#\s*shape:\s*(.+)$", re.M)


def test_counter_figures_quoted_with_a_shape_come_from_a_file_of_that_one_shape():
    """round 4 quoted per-wave counter figures "at rate 1.77" from a file that averaged launches of two shapes.
    Convention from round 5 on: a counter file that DESIGN.md cites from `profiles/r05/` on carries a first line
    `# shape: ...` naming the ONE launch shape all its rows are of, and every sentence that cites it names every number of that
    header (rate, taps, size) -- so the prose cannot drift to another shape than the file's."""
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    checked = 0
    for s in sentences(text):
        for c in CITE.findall(s):
            if "counters" not in os.path.basename(c) or not re.search(r"profiles/r0[5-9]/", c):
                continue
            body = open(os.path.join(ROOT, c), errors="replace").read()
            m = SHAPE_HDR.search(body.split("\n", 1)[0] + "\n")
            assert m, "%s: a counter file cited by DESIGN.md starts with '# shape: ...'" % c
            for num in re.findall(r"\d+(?:\.\d+)?(?:\^\d+)?", m.group(1)):
                assert num in s, "%s is of the shape <%s>; the sentence citing it does not say %s: <<%s>>" % (c, m.group(1), num, s[:200])
            checked += 1
    assert checked >= 1, "DESIGN.md cites no round-5 counter file: the general-rate paragraph (4.3b) should"
