#!/usr/bin/env python
"""Given the stderr of a process that died under LIVO2_REDZONE=2/3 (ROCr's 'Memory access fault ... on address X' + the abort handler's table of
allocations, fast-livo2_amd/csrc/dev_alloc.hpp), print the allocations nearest to the faulting address."""
import re
import sys

txt = open(sys.argv[1], errors="ignore").read()
m = re.search(r"on address (0x[0-9a-f]+)", txt)
if not m:
    raise SystemExit("no fault address in " + sys.argv[1])
addr = int(m.group(1), 16)
rows = []
for mm in re.finditer(r"(freed )?(0x[0-9a-f]+) \.\. (0x[0-9a-f]+)\s+(\d+) B\s+line (\d+)", txt):
    a, b, n, l = int(mm.group(2), 16), int(mm.group(3), 16), int(mm.group(4)), int(mm.group(5))
    d = addr - b if addr >= b else (a - addr if addr < a else 0)
    # the fault address is page-granular: an access anywhere in the 4-KB page counts as inside
    if addr < a and (addr | 0xfff) >= a:
        d = 0
    rows.append((d, ("FREED, " if mm.group(1) else "") + ("behind its end" if addr >= b else ("in front of its start" if (addr | 0xfff) < a else "inside")), a, b, n, l))
rows.sort()
print("fault address", hex(addr), "(page-granular);", len(rows), "live + freed allocations")
for d, side, a, b, n, l in rows[:5]:
    print(f"  {d:>12} B {side:22} of {hex(a)}..{hex(b)} ({n} B) allocated at source part {l // 100000} line {l % 100000} (parts: dev_alloc.hpp src_part)")
