"""stdin: llvm-readelf --notes of a gfx950 code object; stdout: one line per kernel with its register / spill / scratch counts"""
import re, subprocess, sys

txt = sys.stdin.read()
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s*(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    name = re.sub(r"\(anonymous namespace\)::|mp::|void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("false", "0").replace("true", "1").replace(", ", ",")
    rows.append((name, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                 g("group_segment_fixed_size"), g("kernarg_segment_size")))
print(f"{'kernel':52s} {'vgpr':>4s} {'sgpr':>4s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'kernarg':>7s}")
for r in sorted(rows):
    print(f"{r[0][:52]:52s} {r[1]:>4s} {r[2]:>4s} {r[3]:>6s} {r[4]:>6s} {r[5]:>7s} {r[6]:>6s} {r[7]:>7s}")
