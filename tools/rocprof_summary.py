"""Turn a rocprofv3 (rocpd sqlite) result into the plain-text per-kernel summary committed under profiles/."""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats ; durations in microseconds\n")
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for name, calls, tot, avg, pct in rows:
            f.write(f"\"{name}\",{calls},{tot:.1f},{avg:.1f},{pct:.3f}\n")
    print(f"wrote {out} ({len(rows)} kernels)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
