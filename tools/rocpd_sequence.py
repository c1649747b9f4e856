"""The launch sequence of the last step of a rocprofv3 --kernel-trace run (ROCm 7.2 rocpd sqlite): every dispatch with its
queue / stream, start, duration and the gap to the dispatch before on the same queue.  Usage: python tools/rocpd_sequence.py <results.db> [last_ms]"""
import sqlite3
import sys


def main(path, last_ms=30.0):
    con = sqlite3.connect(path)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        print("tables / views:", names)
        return
    cols = [d[0] for d in cur.execute("select * from %s limit 1" % view).description]
    print("columns of %s: %s" % (view, cols))
    def pick(*cands):
        for c in cands:
            if c in cols:
                return c
        return None
    cs, ce, cn = pick("start"), pick("end"), pick("name", "kernel_name")
    cq = pick("queue_id", "queue", "stream_id", "stream")
    cst = pick("stream_id", "stream")
    rows = list(cur.execute("select %s, %s, %s, %s, %s from %s order by %s" % (cs, ce, cn, cq or "0", cst or "0", view, cs)))
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
    last_end = {}
    for s, e, n, q, st in rows:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print("%10.1f us  q %-4s st %-4s  %8.1f us  gap %7.1f  %s" % ((s - rows[0][0]) / 1e3, q, st, (e - s) / 1e3, gap, n[:70]))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
