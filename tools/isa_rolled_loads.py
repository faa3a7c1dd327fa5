"""Kernels in a hipcc -S listing that keep a global load inside a rolled loop which drains the memory queue every trip (a
`s_waitcnt vmcnt(0)` between the load and the back edge): each trip is a full memory round trip.  tools/isa_rolled_loads.py file.s"""
import re, sys
name = None; labels = {}; body = []
def flush():
    if name is None: return
    out = []
    for i, l in enumerate(body):
        m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            loads = sum(1 for s in seg if re.match(r'\s*(global|flat|buffer)_load', s))
            drains = sum(1 for s in seg if 'vmcnt(0)' in s)
            if loads and drains: out.append((m.group(1), len(seg), loads, drains))
    if out: print(name[:150], out)
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        flush(); name = m.group(1); labels = {}; body = []; continue
    if name is None: continue
    m = re.match(r'^(\.LBB\d+_\d+):', line)
    if m: labels[m.group(1)] = len(body)
    body.append(line)
    if 's_endpgm' in line:
        flush(); name = None
