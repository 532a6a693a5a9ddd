"""DESIGN.md's current-state table must quote the round's TRACKED bench line, not an older one (the round-5 verdict found 11.85 ms / 0.50 in the table while
every tracked file said 12.1-13.0 / 0.46-0.49).  The table carries a machine-readable marker

    <!-- headline-of-record: file=profiles/rNN_bench.json mpix_s=... fwd_ms=... bwd_ms=... bwd_frac=... -->

whose numbers must equal the named file's to 5 % and must appear in the table's two headline rows as printed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_headline_rows_quote_the_tracked_bench_line():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- headline-of-record: file=(\S+) mpix_s=([\d.]+) fwd_ms=([\d.]+) bwd_ms=([\d.]+) bwd_frac=([\d.]+) -->", text)
    assert m, "DESIGN.md lost its headline-of-record marker"
    path, mpix, fwd, bwd, frac = m.group(1), *(float(v) for v in m.groups()[1:])
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d\d_bench\.json", f))[-1]
    assert os.path.basename(path) == newest, f"the table quotes {path}, the newest tracked bench line is profiles/{newest}"
    line = json.loads(open(os.path.join(ROOT, path)).read().strip().splitlines()[-1])
    summ = line["summary"]
    for name, got, want in (("Mpix/s", mpix, line["value"]), ("fwd_ms", fwd, summ["fwd_ms"]), ("bwd_ms", bwd, summ["bwd_ms"]), ("bwd_frac", frac, line["roofline"]["frac"])):
        assert abs(got - want) <= 0.05 * want, (name, got, want)
    rows = {k: next(l for l in text.splitlines() if l.startswith(f"| `{k}`")) for k in ("render_fwd2x_k", "render_bwd_pair_k")}
    assert m.group(3) in rows["render_fwd2x_k"] and m.group(4) in rows["render_bwd_pair_k"] and m.group(5) in rows["render_bwd_pair_k"], \
        "the headline rows of the table do not print the marker's numbers"
