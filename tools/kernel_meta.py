"""Register / LDS / spill metadata of the kernels in a hipcc -S listing: tools/kernel_meta.py file.s [name-substring]"""
import re, sys
name, cur = None, {}
want = sys.argv[2] if len(sys.argv) > 2 else ""
keys = ('.vgpr_count', '.sgpr_count', '.sgpr_spill_count', '.vgpr_spill_count', '.group_segment_fixed_size', '.private_segment_fixed_size')
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith('.name:') and '_Z' in l:
        name = l.split(':', 1)[1].strip()
    for k in keys:
        if l.startswith(k + ':'):
            cur[k] = l.split(':')[1].strip()
    if l.startswith('.vgpr_spill_count') and name:
        if want in name:
            print(name[:110], ' '.join('%s=%s' % (k.strip('.').replace('_count', '').replace('_segment_fixed_size', ''), cur.get(k)) for k in keys))
        cur = {}
