"""Compact per-kernel register / spill table of one kernel source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kres.py eve_amd/csrc/ffn_mfma.hip [substring ...] [-- -DEVE_X=1 ...]

Prints one line per kernel whose mangled name contains every substring: VGPRs, AGPRs, SGPRs, scratch bytes (spills),
LDS.  Compile only (no GPU).
"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[: args.index("--")]
src, subs = args[0], args[1:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Iinclude", "-Ieve_amd/csrc",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if all(s in name for s in subs):
        short = re.sub(r"^_ZN3eve\d+", "", name)
        print(f"{short[:70]:70s} v{r.get('VGPRs', -1):4d} a{r.get('AGPRs', -1):4d} s{r.get('TotalSGPRs', -1):4d} "
              f"scratch {r.get('ScratchSize', -1):5d} vspill {r.get('VGPRs Spill', -1):3d} lds {r.get('LDS Size', -1)}")
if not rows:
    print(err[-2000:])
