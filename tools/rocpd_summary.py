"""Turns a rocprofv3 result database (rocpd sqlite, the default output of `rocprofv3 --kernel-trace
--stats`) into the per-kernel stats table committed under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                      "order by total_duration desc").fetchall()
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {calls} | {total / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
