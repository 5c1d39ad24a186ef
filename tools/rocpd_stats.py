"""Per-kernel statistics from a rocprofv3 rocpd (sqlite) kernel trace -> text / CSV summary.
Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:150]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    q = ("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    try:
        rows = list(cur.execute(q))
    except Exception as e:
        print("schema:", cols)
        raise
    stats = {}
    for name, start, end, gx, wx in rows:
        key = short(name)
        st = stats.setdefault(key, [0, 0, 1 << 62, 0])
        d = end - start
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.2f' % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, v[2] / 1e3,
                                                           v[3] / 1e3, 100.0 * v[1] / total))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
