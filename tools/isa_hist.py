"""Instruction histogram of one kernel's gfx950 ISA (no GPU needed): where do an epilogue's VALU slots go?

    python tools/isa_hist.py eve_amd/csrc/ffn_mfma.hip grouped_gemm_pppILb0ELb0ELi2ELi0E --after-last-mfma
    python tools/isa_hist.py eve_amd/csrc/gate.hip gate_wgrad_bf16_kernelILb1E

Compiles the file with the build's flags to assembly (`--cuda-device-only -S`, cached under /tmp by source
mtime), cuts the first kernel whose mangled name contains the given substring at its `.Lfunc_end` label, and counts
opcodes -- over the whole body, or only behind the last MFMA (the E phase of the GEMM kernels).  This is how
the four-op shuffle around every scalar bf16 conversion in the activation epilogue was found.
"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from eve_amd.build import FLAGS  # noqa: E402

src, sub = sys.argv[1], sys.argv[2]
tail_only = "--after-last-mfma" in sys.argv
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
asm = "/tmp/isa_%s_%d.s" % (os.path.basename(src), int(os.path.getmtime(src)))
if not os.path.exists(asm):
    flags = [f for f in FLAGS if f not in ("-Wall",)]
    subprocess.run(["hipcc", *flags, "--cuda-device-only", "-S", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.split(";")[0].rstrip().endswith(":")]
if not starts:
    raise SystemExit(f"no kernel whose name contains {sub!r} in {asm}")
a = starts[0]
e = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))  # not the first s_endpgm: early exits
body = lines[a + 1:e]
if tail_only:
    last = max(i for i, l in enumerate(body) if "v_mfma" in l)
    body = body[last + 1:]
ops = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z]\w+)", l)
    if m:
        ops[m.group(1)] += 1
valu = sum(v for k, v in ops.items() if k.startswith("v_") and "mfma" not in k)
trans = sum(v for k, v in ops.items() if re.match(r"v_(exp|rcp|rsq|sqrt|log|sin|cos)_", k))
print(lines[a].split(":")[0])
print(f"{len(body)} lines; VALU {valu} (of which quarter-rate transcendentals {trans}); MFMA "
      f"{sum(v for k, v in ops.items() if 'mfma' in k)}; LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}; "
      f"global/buffer {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_')))}")
for k, v in ops.most_common(top):
    print(f"{v:6d}  {k}")
