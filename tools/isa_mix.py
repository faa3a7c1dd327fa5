"""Instruction mix of one kernel in a hipcc -S listing: tools/isa_mix.py file.s <mangled-name-prefix>"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = None
for i, l in enumerate(lines):
    if l.startswith(sys.argv[2]) and ':' in l and not l.startswith('\t'):
        start = i
        break
cnt = collections.Counter()
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith('s_endpgm'):
        break
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
        continue
    cnt[s.split()[0]] += 1
print("total", sum(cnt.values()))
groups = collections.Counter()
for k, v in cnt.items():
    g = 'v_f64' if k.startswith('v_') and 'f64' in k else ('v_lane' if 'lane' in k else (k.split('_')[0] + '_' + (k.split('_')[1] if k.startswith(('ds_', 'global_', 'scratch_')) else 'other')))
    groups[g] += v
print(dict(groups))
for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("%6d %s" % (v, k))
