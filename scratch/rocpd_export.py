"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database by default; export the two tables the round's scripts read:
    python scratch/rocpd_export.py stats <results.db> <out.csv>       per-kernel totals (the `--stats` kernel table)
    python scratch/rocpd_export.py counters <results.db> <outdir>     one row per (dispatch, counter), dispatch order ->
                                                                       <outdir>/export_counter_collection.csv
    python scratch/rocpd_export.py trace <results.db> <out.csv>       one row per kernel dispatch (name, start, end, grid)"""
import csv
import os
import sqlite3
import sys


def main():
    mode, db, out = sys.argv[1], sys.argv[2], sys.argv[3]
    c = sqlite3.connect(db)
    if mode == "stats":
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                         "order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows)
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for n, k, t, a, mn, mx in rows:
                w.writerow([n, k, int(t), round(a, 1), round(100.0 * t / tot, 3), int(mn), int(mx)])
    elif mode == "counters":
        os.makedirs(out, exist_ok=True)
        rows = c.execute("select dispatch_id, kernel_name, counter_name, value, grid_size, workgroup_size, start, end from counters_collection "
                         "order by dispatch_id").fetchall()
        with open(os.path.join(out, "export_counter_collection.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "Start_Timestamp", "End_Timestamp"])
            w.writerows(rows)
    elif mode == "trace":
        rows = c.execute("select dispatch_id, name, start, end, duration, grid_x, grid_y, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "DurationNs", "Grid_X", "Grid_Y", "Workgroup_X", "VGPR", "LDS"])
            w.writerows(rows)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
