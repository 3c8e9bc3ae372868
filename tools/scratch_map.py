#!/usr/bin/env python3
"""Developer tool: where (between which barriers) a kernel of /tmp/mvfit_api.s touches scratch. usage: scratch_map.py <mangled-prefix>"""
import sys, collections
name = sys.argv[1] if len(sys.argv) > 1 else '_ZN5mvfit21fit_persistent_kernelILb0ELb0ELb1E'
lines = open('/tmp/mvfit_api.s').read().split('\n')
a = next(i for i, l in enumerate(lines) if l.startswith(name))
b = next(i for i in range(a, len(lines)) if lines[i].startswith('\t.amdhsa_kernel ' + name))
body = [l for l in lines[a:b] if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
bar = [i for i, l in enumerate(body) if 's_barrier' in l]
seg = collections.Counter()
for i, l in enumerate(body):
    if 'scratch_' in l:
        seg[(sum(1 for x in bar if x < i), 's' if 'store' in l else 'l')] += 1
calls = [sum(1 for x in bar if x < i) for i, l in enumerate(body) if 's_swappc' in l]
print('instrs', len(body), 'barriers at', bar)
print('calls in segments', calls)
print(' '.join('%d%s:%d' % (k[0], k[1], v) for k, v in sorted(seg.items())))
