"""Turn rocprofv3 SQLite outputs into small text/CSV summaries (for profiles/).

    python tools/export_profile.py <stats_dir> [<pmc_dir> ...]
Prints a kernel table (calls, total, average µs, %) for the first directory and, for
every further directory, the per-kernel average of each collected PMC counter.
"""
import glob
import sqlite3
import sys


def db_in(d):
    hits = glob.glob(f"{d}/**/*.db", recursive=True)
    return hits[0] if hits else None


def short(name, n=90):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


# kernels of fm_points.hip that sit outside namespace fm (the softmin sweep)
OURS_OUTSIDE_NAMESPACE = ("softmin_", "random_subset", "random_state")


def kernel_table(path):
    """This package's kernels (namespace fm) one per row; everything else — torch's elementwise / GEMM kernels of the
    synthetic-scene construction before the timed region — summed into one row."""
    con = sqlite3.connect(path)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("kernel,calls,total_us,avg_us,percent")
    other = [0, 0.0, 0.0]
    for name, calls, total, avg, pct in rows:
        if "fm::" in name or name.startswith(OURS_OUTSIDE_NAMESPACE):
            print(f"\"{short(name)}\",{calls},{total:.1f},{avg:.2f},{pct:.2f}")
        else:
            other[0] += calls
            other[1] += total
            other[2] += pct
    if other[0]:
        print(f"\"(not fm::) torch kernels of the scene / track synthesis and set-up, outside the timed steps\",{other[0]},{other[1]:.1f},"
              f"{other[1] / other[0]:.2f},{other[2]:.2f}")


def pmc_table(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    print("# columns:", cols)
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    cnt_col = "counter_name" if "counter_name" in cols else None
    val_col = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
    if not (name_col and cnt_col and val_col):
        for r in con.execute("select * from counters_collection limit 5"):
            print(r)
        return
    q = f"select {name_col}, {cnt_col}, count(*), avg({val_col}), sum({val_col}) from counters_collection group by {name_col}, {cnt_col} order by sum({val_col}) desc limit 25"
    print("kernel,counter,dispatches,avg_value,sum_value")
    for name, cnt, n, avg, tot in con.execute(q):
        print(f"\"{short(name)}\",{cnt},{n},{avg:.1f},{tot:.1f}")


if __name__ == "__main__":
    dirs = [a for a in sys.argv[1:] if a != "--pmc-only"]
    pmc_only = "--pmc-only" in sys.argv  # every directory is a PMC pass (no --stats pass first)
    for i, d in enumerate(dirs):
        path = db_in(d)
        print(f"## {d}: {path}")
        if not path:
            continue
        try:
            if i == 0 and not pmc_only:
                kernel_table(path)
            else:
                pmc_table(path)
        except Exception as exc:  # keep going: the raw db is merged back anyway
            print("error:", exc)
