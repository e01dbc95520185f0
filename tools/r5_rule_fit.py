#!/usr/bin/env python3
"""Round 5: joins tools/r5_rule_sweep.sh's measurements with plan()'s packing model (restated here for the two-wave kernels) -- per launch shape: the model's price
of the strips (of at most 512 rows: what the balanced rule compares with) over one round of chunks, the chunk's rows and how it sits in the strip column, what plan() does,
and the measured gain of the chunks over the strips' default form.   usage: python3 tools/r5_rule_fit.py gpurun_out/<dir>/sweep.txt"""
import re
import sys


def cell_rows_for(h):
    return 32 if h >= 2048 else 8


def model(w, h, n, slots=2048, simds=1024):
    cr = cell_rows_for(h)
    strips_x = (w + 127) // 128
    rc = lambda v: (v + cr - 1) // cr * cr
    tail2 = (685, 1000)
    cap = 1024
    best, best_rows, best_capped = None, None, None
    ny_min, ny_max = (h + cap - 1) // cap, (h + cr - 1) // cr
    for ny in range(ny_min, ny_max + 1):
        rows = rc((h + ny - 1) // ny)
        if (h + rows - 1) // rows != ny:
            continue
        cnt, u = strips_x * ny * n, rows + 12
        full, rem = divmod(cnt, slots)
        last = 0 if rem == 0 else tail2[(rem - 1) // simds]
        cost = u * (full * 1000 + last)
        if best is None or cost < best:
            best, best_rows = cost, rows
        if rows <= 512 and (best_capped is None or cost < best_capped):
            best_capped = cost
    col_cells = (h + cr - 1) // cr
    allc = n * strips_x * col_cells
    if allc <= slots:
        return None
    chunk = (allc + slots - 1) // slots
    n_chunks = (allc + chunk - 1) // chunk
    tail = 1000 if n_chunks > simds else 685
    chunks_cost = (chunk * cr * col_cells + 12 * (col_cells + chunk)) * tail / col_cells
    strips_n = strips_x * ((h + best_rows - 1) // best_rows) * n
    return dict(cr=cr, col_cells=col_cells, chunk=chunk, chunk_rows=chunk * cr, ratio_capped=best_capped / chunks_cost, ratio_best=best / chunks_cost,
                rows=best_rows, strips=strips_n, even=col_cells % chunk == 0)


res = {}
cur = None
for l in open(sys.argv[1]):
    m = re.match(r"lib \S+ \| pairs (\d+) size (\d+)x(\d+) mode", l)
    if m:
        cur = (int(m.group(2)), int(m.group(3)), int(m.group(1)))
        res[cur] = {}
    m = re.match(r"\s+variant (\d): median \S+ ms \(([\d.]+) Gpix", l)
    if m:
        res[cur][int(m.group(1))] = float(m.group(2))
print("# size x pairs | strips plan() picks | chunk rows (cells; of the column's) | model: strips<=512 / chunks, best strips / chunks | Gpix/s: strips (2 | 3) chunks (6) default (0) | chunks vs the strips' default form")
for (w, h, n), v in res.items():
    md = model(w, h, n)
    if md is None or 6 not in v:
        continue
    early = md["strips"] <= 3 * 2048
    strips = v[3] if early else v[2]
    print("%4dx%-4d x %3d | %5d strips of %4d rows | %4d rows (%3d of %3d cells%s) | %.3f %.3f | %6.1f %6.1f  %6.1f  %6.1f | %+5.1f %%"
          % (w, h, n, md["strips"], md["rows"], md["chunk_rows"], md["chunk"], md["col_cells"], ", even" if md["even"] else "", md["ratio_capped"], md["ratio_best"], v[2], v[3], v[6], v[0], (v[6] / strips - 1) * 100))
