"""Summarise a rocprofv3 (--kernel-trace --stats) sqlite result into a per-kernel table (markdown)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
span = db.execute("select min(start), max(end) from kernels").fetchone()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:110]
print(f"total kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} launches; first-to-last kernel span {(span[1]-span[0])/1e6:.1f} ms\n")
print("| kernel | calls | total ms | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|")
for n, c, s, a, mn, mx in rows[:40]:
    print(f"| {short(n)} | {c} | {s/1e6:.2f} | {100*s/tot:.1f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} |")
