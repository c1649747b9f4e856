"""Text summary of rocprofv3 (ROCm 7.2 rocpd sqlite) outputs: kernel-trace stats and
per-kernel PMC counter sums.  Usage: python tools/rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def main(paths):
    for path in paths:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print("== %s" % path)
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print("%-90s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for name, calls, tot, avg, pct in rows:
                print("%-90s %6d %12.1f %12.2f %6.2f%%" % (name[:90], calls, tot / 1e3 if tot > 1e6 else tot,
                                                          avg / 1e3 if tot > 1e6 else avg, pct))
        try:
            q = ("select kernel_name, counter_name, count(distinct dispatch_id), sum(value), "
                 "avg(vgpr_count), avg(lds_block_size) from counters_collection group by kernel_name, counter_name")
            pm = list(cur.execute(q))
        except sqlite3.Error:
            pm = []
        if pm:
            print("%-60s %-28s %9s %18s %16s" % ("kernel", "counter", "dispatch", "sum", "per-dispatch"))
            for kn, cn, nd, sv, vg, lds in pm:
                if any(t in kn for t in ("conv_mfma", "kernel", "pw_gemm", "pw_head", "wino")):
                    print("%-60s %-28s %9d %18.0f %16.1f" % (kn[:60], cn, nd, sv, sv / max(nd, 1)))
        con.close()


if __name__ == "__main__":
    main(sys.argv[1:])
