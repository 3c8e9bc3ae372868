#!/bin/bash
# Developer tool: emit the gfx950 ISA of mvfit_api.hip and print, per fit kernel, registers / scratch / code size and
# instruction-class counts (the numbers DESIGN.md quotes: .private_segment_fixed_size, spills, barriers).
# usage: tools/isa_stats.sh [extra hipcc flags]     -> /tmp/mvfit_api.s
set -e
cd "$(dirname "$0")/../mvsmplfitting_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -I../../include \
    --cuda-device-only -S mvfit_api.hip -o /tmp/mvfit_api.s "$@" 2>/dev/null
python3 - <<'PY'
import re
txt = open('/tmp/mvfit_api.s').read()
meta = re.findall(r'\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n){0,4}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', txt)
lines = txt.split('\n')
for name, priv, sg, sgs, vg, vgs in meta:
    if 'fit_persistent' not in name and 'fit_step' not in name and 'closure_kernel' not in name:
        continue
    try:
        a = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
        b = next(i for i in range(a, len(lines)) if lines[i].startswith('\t.amdhsa_kernel ' + name))
    except StopIteration:
        continue
    body = lines[a:b]
    cnt = lambda pat: sum(1 for l in body if re.search(pat, l))
    ins = sum(1 for l in body if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'))
    short = re.sub(r'_ZN5mvfit\d+', '', name)[:44]
    print('%-44s vgpr %3s sgpr-spill %3s vgpr-spill %3s scratch %4s B | instr %5d barriers %2d scratch ld/st %3d/%3d global ld %3d lds r/w %3d/%3d f64 %3d exec-branch %3d'
          % (short, vg, sgs, vgs, priv, ins, cnt(r's_barrier'), cnt(r'scratch_load'), cnt(r'scratch_store'), cnt(r'global_load'),
             cnt(r'ds_read|ds_load'), cnt(r'ds_write|ds_store'), cnt(r'_f64'), cnt(r's_cbranch_exec')))
PY
