#!/usr/bin/env python3
"""profiles/r0N_instruction_budget.md from two rounds' profile summaries:
    python tools/instruction_budget.py profiles/r04 profiles/r05 <steps traced> > profiles/r05_instruction_budget.md
SQ_INSTS_VALU per dispatch (r0N_pmc.json: counters) x dispatches per step (r0N_pmc.json: kernel_trace calls / traced steps)."""
import json
import re
import sys

a, b = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 9.0  # bench.py --steps 5 --warmup 1 + 3 event-timed steps
skip = re.compile(r"synth_kernel|rocclr|vectorized_elementwise|finite_kernel")


def load(prefix):
    d = json.load(open(prefix + "_pmc.json"))
    out = {}
    for k, tr in d["kernel_trace"].items():
        if skip.search(k) or k not in d["counters"]:
            continue
        out[k] = (d["counters"][k].get("SQ_INSTS_VALU", 0.0), tr["calls"] / steps, tr["avg_us"])
    return out, d["source_sha256"]


A, ha = load(a)
B, hb = load(b)
ra, rb = a[-3:], b[-3:]
names = sorted(set(A) | set(B), key=lambda k: -max(A.get(k, (0, 0, 0))[0] * A.get(k, (0, 0, 0))[1], B.get(k, (0, 0, 0))[0] * B.get(k, (0, 0, 0))[1]))
sa = sum(v[0] * v[1] for v in A.values())
sb = sum(v[0] * v[1] for v in B.values())
print(f"| kernel | {ra} instr / dispatch | {ra} dispatches / step | {ra} instr / step | {ra} avg us | {rb} instr / dispatch | {rb} dispatches / step | {rb} instr / step | {rb} avg us |")
print("|---|---|---|---|---|---|---|---|---|")
for k in names:
    cells = []
    for T in (A, B):
        if k in T:
            i, n, us = T[k]
            cells += [f"{i:.3e}", f"{n:.2f}", f"{i * n:.3e}", f"{us:.1f}"]
        else:
            cells += ["", "", "", ""]
    print(f"| {k} | " + " | ".join(cells) + " |")
print(f"| **sum** | | | {sa:.3e} | | | | {sb:.3e} | |")
print(f"\n(sources: `{a}_pmc.json` hash {ha}, `{b}_pmc.json` hash {hb}; {steps:g} steps traced)", file=sys.stderr)
