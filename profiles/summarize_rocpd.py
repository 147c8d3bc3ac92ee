"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average, like `--stats`.
usage: python profiles/summarize_rocpd.py <results.db> [out.md]"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" % (short(n), c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    lines.append("\ntotal kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
