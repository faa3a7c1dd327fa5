"""Per-barrier-segment instruction mix of one kernel in a hipcc -S listing: tools/isa_segments.py file.s <mangled-name-prefix>"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l and not l.startswith('\t'))
seg, segs = collections.Counter(), []
def cls(k):
    if k.startswith('v_') and 'f64' in k: return 'f64'
    if 'lane' in k: return 'lane'
    if k.startswith('v_'): return 'valu'
    if k.startswith('s_load') or k.startswith('s_buffer'): return 'smem'
    if k.startswith('s_waitcnt'): return 'wait'
    if k.startswith('s_cbranch') or k.startswith('s_branch'): return 'br'
    if k.startswith('s_'): return 'salu'
    if k.startswith('ds_read') or k.startswith('ds_load'): return 'dsr'
    if k.startswith('ds_'): return 'dsw'
    if k.startswith('global_load') or k.startswith('buffer_load'): return 'vld'
    if k.startswith('global_') or k.startswith('buffer_'): return 'vst'
    if k.startswith('scratch'): return 'scr'
    return 'other'
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith('s_endpgm'): break
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
    k = s.split()[0]
    if k == 's_barrier':
        segs.append(seg); seg = collections.Counter(); continue
    seg[cls(k)] += 1
segs.append(seg)
keys = ['f64', 'valu', 'lane', 'salu', 'smem', 'wait', 'br', 'dsr', 'dsw', 'vld', 'vst', 'scr']
print('seg  ' + ' '.join('%5s' % k for k in keys))
for i, s in enumerate(segs):
    print('%3d  ' % i + ' '.join('%5d' % s[k] for k in keys))
tot = sum(segs, collections.Counter())
print('all  ' + ' '.join('%5d' % tot[k] for k in keys))
