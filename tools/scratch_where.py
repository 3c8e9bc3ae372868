"""Developer tool: where the spill code of a fit kernel sits (tools/isa_stats.sh writes /tmp/mvfit_api.s first).
usage: python tools/scratch_where.py [substring of the mangled kernel name, default the lean single-launch kernel]"""
import re
import sys

sub = sys.argv[1] if len(sys.argv) > 1 else 'fit_persistent_kernelILb0ELb0ELb1ELb0'
lines = open('/tmp/mvfit_api.s').read().split('\n')
a = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and sub in l and ': ' in l and not l.startswith('\t'))
name = lines[a].split(':')[0]
b = next(i for i in range(a, len(lines)) if lines[i].startswith('\t.amdhsa_kernel ' + name))
body = lines[a:b]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
back = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            back.append((labels[t], i))
big = [(x, y) for x, y in back if y - x > 3000]
print(name[:60], 'lines', len(body), 'round loop', big[:3])
inl = lambda i: any(x <= i <= y for x, y in big)
for i, l in enumerate(body):
    if 'scratch_' in l:
        print(i, l.strip(), '  <-- IN THE ROUND LOOP' if inl(i) else '')
print('in the round loop: v_writelane %d, v_readlane %d (SGPR spills to VGPR lanes), scratch %d'
      % (sum('v_writelane' in l and inl(i) for i, l in enumerate(body)), sum('v_readlane' in l and inl(i) for i, l in enumerate(body)),
         sum('scratch_' in l and inl(i) for i, l in enumerate(body))))
