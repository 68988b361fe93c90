"""Instruction mix of the MFMA loops of one kernel in a hipcc -S listing: loopmix.py file.s <kernel-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
start = [i for i, l in enumerate(lines) if sub in l and l.rstrip().split(':')[0].startswith('_Z') and ': ' in l and l.startswith('_Z')][0]
end = next(i for i in range(start, len(lines)) if '.amdhsa_kernel' in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        seg = body[labels[mm.group(1)]:i + 1]
        if sum('v_mfma' in x for x in seg) >= 16:
            c = collections.Counter()
            for x in seg:
                x = x.strip()
                if not x or x.startswith(';') or x.startswith('.'):
                    continue
                op = x.split()[0]
                c['mfma' if 'mfma' in op else 'ds_read' if op.startswith('ds_read') else 'ds_write' if op.startswith('ds_write') else
                  'vmem_load' if op.startswith('buffer_load') or op.startswith('global_load') else 'vmem_store' if 'store' in op else
                  op if op in ('s_waitcnt', 's_barrier', 's_nop') else 'valu' if op.startswith('v_') else 'salu'] += 1
            print("loop", mm.group(1), "lines", len(seg), dict(c))
            if len(sys.argv) > 3:
                open(sys.argv[3] + mm.group(1).strip('.') + '.s', 'w').write('\n'.join(seg))
