"""HBM-side traffic of the LBS vertex pass from rocprofv3 PMC counters -> profiles/<tag>_pmc.json (read by bench.py).

Two roles:
  * `python tools/pmc_vertex_pass.py drive B [launches]`  - the workload the counters are collected on: `launches`
    vertex passes over B problems (mvfit_vertices), nothing else of interest;
  * `python tools/pmc_vertex_pass.py parse <out.json> <name>=<dir> ...` - reads the *_counter_collection.csv files of
    the passes (separate rocprofv3 runs per counter, as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and
    WRITE_SIZE do not fit one pass) and writes per kernel: launches, FETCH_SIZE / WRITE_SIZE averages (KiB),
    traffic = 2 x FETCH_SIZE (gfx950 reports half of a 16-byte-per-lane stream) + WRITE_SIZE.

A third pass (`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES`) gives the matrix-pipe utilisation.

On the GPU box (see tools/collect_profiles.sh):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/b32_fetch -o p -- python tools/pmc_vertex_pass.py drive 32
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def drive(B, launches=40):
    import numpy as np
    import torch
    from mvsmplfitting_amd import synthetic as syn
    from mvsmplfitting_amd.engine import MvFit
    eng = MvFit(syn.make_body_model(0, skin_topk=4))
    cams = syn.make_camera_ring(8)
    rng = np.random.default_rng(B)
    x = np.zeros((B, 118), np.float32)
    x[:, :86] = rng.normal(0, 0.2, (B, 86))
    x[:, 85] = 1.0
    eng.set_problems(cams, np.zeros((B, 8, 17, 2), np.float32), np.ones((B, 8, 17), np.float32))
    for _ in range(launches):
        eng.vertices(x)
    torch.cuda.synchronize()
    eng.close()


def drive_resident(B, side_file, reps=3, rounds=100):
    """The RESIDENT pass as the counters' workload: one asynchronous fit of B problems (its resident launch serves every
    closure round of the fit) + `reps` stand-alone resident launches over `rounds` rounds of the ring the fit left behind.
    Every dispatch of lbs_vertex_pass_resident_kernel is summed by parse(); the number of closure rounds they served goes
    to `side_file` so that the traffic can be stated per round."""
    import numpy as np
    import torch
    from mvsmplfitting_amd import synthetic as syn
    from mvsmplfitting_amd.engine import MvFit, stage_weights
    import bench
    eng = MvFit(syn.make_body_model(0, skin_topk=4))
    cams, gt, conf, x0 = bench.build_inputs(eng, syn, 0, B, 1, 8)
    xf, st = eng.fit(x0, stage_weights(1536.0))
    pp = eng.pass_profile()
    served = int(st['n_closure'].max().item()) if pp['tiles_per_workgroup'] else 0
    dispatches = 1 if pp['tiles_per_workgroup'] else 0
    if pp['tiles_per_workgroup']:
        for _ in range(reps):
            eng.profile_resident_pass_ms(rounds)
            served += rounds
            dispatches += 1
    torch.cuda.synchronize()
    with open(side_file, 'w') as f:
        json.dump(dict(rounds_served=served, dispatches=dispatches, tiles_per_workgroup=pp['tiles_per_workgroup'], problems=B,
                       standalone_dispatches=reps if pp['tiles_per_workgroup'] else 0, rounds_per_standalone_dispatch=rounds), f)
    eng.close()


def resident_per_round(d, counter, rounds_per_dispatch):
    """Per-round value of `counter` over the STAND-ALONE dispatches of the resident pass in a drive_resident run: the rows of
    lbs_vertex_pass_resident_kernel in dispatch order, the first one dropped (the fit's own dispatch: rocprofv3 serialises
    kernels while it collects counters, so that dispatch cannot run beside its optimiser kernel and serves nothing useful),
    summed over the agents / dimensions rocprofv3 splits a counter into, divided by the rounds the stand-alone dispatches
    served.  -> (value per round, dispatches used) or (None, 0)."""
    per = {}
    for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if 'vertex_pass_resident' in row['Kernel_Name'] and row['Counter_Name'] == counter:
                    per[int(row['Dispatch_Id'])] = per.get(int(row['Dispatch_Id']), 0.0) + float(row['Counter_Value'])
    ids = sorted(per)[1:]
    if not ids:
        return None, 0
    return sum(per[i] for i in ids) / (len(ids) * rounds_per_dispatch), len(ids)


def parse(out_json, specs):
    res = {}
    for spec in specs:
        name, d = spec.split('=', 1)
        acc = {}
        for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k = row['Kernel_Name']
                    if 'vertex_pass' not in k:
                        continue
                    key = (k.split('(')[0].replace('void ', ''), row['Counter_Name'])
                    a = acc.setdefault(key, [0, 0.0])
                    a[0] += 1
                    a[1] += float(row['Counter_Value'])
        for (k, cn), (n, tot) in acc.items():
            e = res.setdefault(name, {}).setdefault(k, {})
            e[cn + ('_KiB_avg' if cn in ('FETCH_SIZE', 'WRITE_SIZE') else '_avg')] = tot / n
            e['launches_' + cn] = n
    for name, ks in res.items():
        for k, e in ks.items():
            if 'FETCH_SIZE_KiB_avg' in e and 'WRITE_SIZE_KiB_avg' in e:
                e['traffic_bytes'] = (2.0 * e['FETCH_SIZE_KiB_avg'] + e['WRITE_SIZE_KiB_avg']) * 1024.0
                e['traffic_note'] = '2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, separate --pmc passes'
            if 'SQ_VALU_MFMA_BUSY_CYCLES_avg' in e:
                # cycles the matrix pipes were busy, summed over all SIMDs of the chip (32 per v_mfma_f32_32x32x16_f16);
                # bench.py divides by launch duration x 2.4 GHz x 1024 SIMDs for the utilisation
                e['mfma_busy_cycles_per_launch'] = e['SQ_VALU_MFMA_BUSY_CYCLES_avg']
    if out_json:
        with open(out_json, 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print(json.dumps(res, indent=1, sort_keys=True))
    return res


if __name__ == '__main__':
    if sys.argv[1] == 'drive':
        drive(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    elif sys.argv[1] == 'drive_resident':
        drive_resident(int(sys.argv[2]), sys.argv[3])
    else:
        parse(sys.argv[2], sys.argv[3:])
