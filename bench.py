#!/usr/bin/env python
"""bench.py - closures/s of the device-resident multi-view SMPL fit on MI355X.

Metric (BASELINE.json): L-BFGS closure evaluations per second (forward + backward, summed over all
concurrently fitted problems) for 8-view, 1-person problems; ms to convergence per frame; the LBS
vertex pass as a fraction of the HBM roofline.

A "step" = ONE complete 4-stage fit (reference cfg_files/fit_smpl.yaml weights, L-BFGS lr=1,
max_iter=30, history=100, strong-Wolfe; outer maxiters=30, ftol=gtol=1e-9) of this rank's batch of
synthetic problems, every closure with its full 6890-vertex LBS pass like the reference's
return_verts=True.  Inputs are resident in HBM before the timed region.  Ranks fit disjoint frames
(weak scaling: the per-GPU share is fixed; --strong: a fixed total split over the ranks); the only
collective is the final all_gather of the fitted parameters (RCCL).

Named workloads (BASELINE.json configs):
  --config configs1   (default) 1 person x 8 views x 32 frames per GPU, GMoF + pose prior, no SDF
  --config demo       configs[0]: the reference's shipped demo inputs (1 frame, 6 real views, VPoser checkpoint)
  --config configs2   configs[1] + the SDF interpenetration term (--sdf-faces wired | all)
  --config configs3   4 persons x 8 views x 256 frames frame-sharded over 8 GPUs = 4 x 32 frames (128 problems) per GPU
  --config configs4   16-view rig x 1024 frames over 8 GPUs = 128 frames per GPU, half-width blendshape operands

  python bench.py --gpus 1 --steps 5 --warmup 1
  python bench.py --gpus 8                       # starts 8 ranks itself (torch.distributed.run, 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CONST_BYTES = 82680 + 826800 + 17114760 + 661440      # v_template + shapedirs + posedirs + lbs_weights
PER_PROBLEM_BYTES = 2032 + 82680                      # (betas, pose_feature, A, transl) in + vertices out
PMC_JSON = os.path.join(ROOT, 'profiles', 'r6_pmc.json')     # written by tools/pmc_vertex_pass.py from rocprofv3 --pmc passes
# the imported reference itself (PyTorch CPU, create_fitting_closure + LBFGSLs + run_fitting) timed in the survey's
# build container (it cannot travel to the GPU box): closures/s inside L-BFGS, 8 vCPUs
SURVEY_REFERENCE_CLOSURES_PER_S = (74.0, 125.0)

PRESETS = {
    'configs1': dict(frames=32, views=8, persons=1),
    'configs2': dict(frames=32, views=8, persons=1, sdf=True),
    'configs3': dict(frames=32, views=8, persons=4),
    'configs4': dict(frames=128, views=16, persons=1, half_basis=True),
}


def bytes_fwd(B, skin_topk=0, half_basis=False):
    """Algorithmic bytes of one LBS vertex pass over B problems; with
    k-sparse skinning weights the weight matrix is k (weight, joint) pairs per vertex instead of 24 floats; the
    half-width basis (configs[4]) is 2 bytes per posedirs / shapedirs element."""
    const = CONST_BYTES if not skin_topk else CONST_BYTES - 661440 + 6890 * skin_topk * 8
    if half_basis:
        const -= 8557380 + 413400
    return const + PER_PROBLEM_BYTES * B


def pmc_file():
    for fn in (PMC_JSON, os.path.join(ROOT, 'profiles', 'r5_pmc.json')):
        if os.path.isfile(fn):
            return fn
    return None


def pmc_value(key, kernel_substr, field='traffic_bytes', table=None):
    """A per-launch figure of a PMC summary (measured in this run, or the committed profiles/*_pmc.json), or None."""
    try:
        if table is None:
            with open(pmc_file()) as f:
                table = json.load(f)
        for k, e in table.get(key, {}).items():
            if kernel_substr in k and field in e:
                return round(e[field])
    except Exception:
        pass
    return None


def measure_pmc(B, launches=40, timeout_s=90):
    """HBM-side traffic and matrix-pipe busy cycles of the vertex pass at B problems, measured NOW with rocprofv3 PMC
    counters on this GPU: separate passes for FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES (the guide's
    recipe: the two size counters do not fit one pass), `--kernel-trace` only, on tools/pmc_vertex_pass.py's driver.
    Returns the parsed table ({'B<n>': {kernel: {...}}}) or None when rocprofv3 is absent or a pass fails."""
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    from tools import pmc_vertex_pass as pv
    out = tempfile.mkdtemp(prefix='mvfit_pmc_')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    specs = []
    try:
        for cn in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            d = os.path.join(out, cn)
            cmd = [exe, '--pmc', cn, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--', sys.executable,
                   os.path.join(ROOT, 'tools', 'pmc_vertex_pass.py'), 'drive', str(B), str(launches)]
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            specs.append('B%d=%s' % (B, d))
        return pv.parse(None, specs)
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_pmc_resident(B, timeout_s=120):
    """The same three PMC passes on the RESIDENT pass (tools/pmc_vertex_pass.py drive_resident): after one fit has filled the
    ring, three stand-alone resident launches serve 100 rounds each; their counters (the fit's own resident dispatch is left
    out: rocprofv3 serialises kernels while collecting, so it cannot run beside its optimiser kernel) are summed and divided
    by the rounds served.  Returns dict(traffic_bytes_per_round, mfma_busy_cycles_per_round, rounds_served, dispatches) or None."""
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    from tools import pmc_vertex_pass as pv
    out = tempfile.mkdtemp(prefix='mvfit_pmc_')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    tot = {}
    used = rounds = 0
    try:
        for cn in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            d = os.path.join(out, cn)
            sf = os.path.join(out, cn + '.json')
            cmd = [exe, '--pmc', cn, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--', sys.executable,
                   os.path.join(ROOT, 'tools', 'pmc_vertex_pass.py'), 'drive_resident', str(B), sf]
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            with open(sf) as f:
                side = json.load(f)
            if not side.get('standalone_dispatches'):
                return None
            rounds = side['rounds_per_standalone_dispatch']
            tot[cn], used = pv.resident_per_round(d, cn, rounds)
            if tot[cn] is None:
                return None
        return dict(traffic_bytes_per_round=round((2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0),
                    mfma_busy_cycles_per_round=tot['SQ_VALU_MFMA_BUSY_CYCLES'], rounds_served=used * rounds, dispatches=used)
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


SHADER_CLOCK_GHZ = 2.4           # MI355X_MICROARCH.md peak engine clock (the resident kernels were measured at ~2.0-2.05 GHz effective)
FP32_VALU_TFLOPS = 157.3         # vector fp32 peak of the chip; per CU: / 256


def true_bound(B, form, workgroups, half_basis=False):
    """What a closure round of the RESIDENT vertex pass cannot be faster than, whatever the schedule (last review, item 1c):
    the largest of
      hbm    the bytes a round really moves (SURVEY 8(d)'s per-problem figure: operands in + 82,680 B of vertices out per problem;
             the 18.2 MB of constants are read once per FIT) / 8 TB/s;
      mfma   the matrix-pipe issue time of the busiest SIMD: 32 cycles per v_mfma_f32_32x32x16_f16, 21 per contraction chain
             (14 with the half-width basis); form 1 (one tile per workgroup) puts 6 chains on 4 SIMDs - two on the busiest -,
             form 3 (two tiles, role-split) 3 chains on every SIMD; per 32-problem chunk;
      valu   the pass's fp32 vector work (skinning blend 4 x 12 FMAs, K halves combined, T applied, "+ transl": 130 flop per
             vertex and problem) / the fp32 vector peak of the CUs the pass runs on.
    Returns dict(us, which, components_us)."""
    nch = -(-B // 32)
    per_chain = 14 if half_basis else 21
    chains_busiest = 2 if form == 1 else 3
    comp = dict(hbm=PER_PROBLEM_BYTES * B / (HBM_PEAK_GBS * 1e9) * 1e6,
                mfma=nch * chains_busiest * per_chain * 32 / (SHADER_CLOCK_GHZ * 1e3),
                valu=6890 * B * 130 / (FP32_VALU_TFLOPS * 1e12 * max(workgroups, 1) / 256) * 1e6)
    which = max(comp, key=comp.get)
    return dict(us=comp[which], which=which, components_us={k: round(v, 3) for k, v in comp.items()})


def resident_round_us(pp, round_period_us):
    """The "launch duration" of a closure round of the resident pass, from the stamps of one profiled fit: the round's service
    SPAN - last workgroup's stores acknowledged minus first workgroup saw the operands.  Always the span (round 6; the earlier
    lines switched to the slowest workgroup's own service time when the pass was the slower side - two figures under one key);
    the slowest workgroup's time is reported next to it (`slowest_workgroup_us`, `frac_slowest_workgroup`), and `pass_keeps_up`
    says whether a round is served before the optimiser publishes the next (else the workgroups drift apart by up to the ring's
    depth, rounds overlap and the span overstates the pass).  Returns (us, 'span')."""
    return pp['round_span_ms'] * 1e3, 'span'


def resident_roofline(eng, B, nbytes, pr, pp, passes, with_pmc, round_period_us=None, half_basis=False):
    """roofline object of the RESIDENT vertex pass (one launch per fit; csrc/vertex_pass.hip): the "launch duration" of a
    closure round is its service span stamped inside the kernel during one complete profiled fit - last workgroup's vertex
    stores acknowledged minus first workgroup saw the round's operands (wall clock, 10 ns) - because a per-round launch no
    longer exists; next to it the pass ALONE (HIP events around one resident launch that serves 100 rounds back to back:
    the figure rocprofv3 --kernel-trace shows as that dispatch's duration / 100)."""
    us, which = resident_round_us(pp, round_period_us)
    span_ms = us * 1e-3
    alone_ms = min(eng.profile_resident_pass_ms(100) for _ in range(3))
    ach = nbytes / (span_ms * 1e-3) / 1e9
    # what a round still has to move: the vertices out + the round's operands in (read by every workgroup from L2 / MALL,
    # once from HBM) - SURVEY 8(d)'s per-problem figure; the 18.2 MB of constants cross the memory system once per FIT
    moved = PER_PROBLEM_BYTES * B
    table = measure_pmc_resident(B) if with_pmc else None
    tb = true_bound(B, pp['form'], pp['workgroups'], half_basis)
    slow_us = pp['slowest_workgroup_ms'] * 1e3
    roof = dict(bound='hbm', kernel=pp['kernel'],
                achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4),
                frac_basis='SURVEY 8(d) algorithmic bytes per LAUNCH (constants re-read every round) / avg_launch_us / 8 TB/s - the '
                           'figure of the earlier rounds, kept for continuity; it is NOT the HBM utilisation of the resident pass, '
                           'which reads the constants once per fit: see bound_true / frac_true and frac_of_peak_for_bytes_moved',
                # the honest figure: the largest hardware floor of a round / the measured round
                bound_true=tb['which'], bound_true_us=round(tb['us'], 3), bound_true_components_us=tb['components_us'],
                frac_true=round(tb['us'] / (span_ms * 1e3), 4),
                bound_true_note='max(bytes moved per round / 8 TB/s, matrix-pipe issue time of the busiest SIMD at %.1f GHz, fp32 '
                                'vector flops / peak of the %d CUs the pass holds); the rest of the round is latency: the operands '
                                'cannot be requested before the tags flip (poll + sc1 LDS-DMA ~1.5 us), one dependent MFMA chain, '
                                'the CU write path (DESIGN 4.1)' % (SHADER_CLOCK_GHZ, pp['workgroups']),
                algorithmic_bytes=nbytes, avg_launch_us=round(span_ms * 1e3, 2), avg_launch_is=which,
                avg_launch_note='resident pass: no per-round launch exists; this is the in-fit service span of a closure round '
                                '(max over workgroups of stores acknowledged - min over workgroups of operands seen), mean over '
                                'the %d rounds of one complete profiled fit' % pp['rounds_stamped'],
                pass_keeps_up=None if round_period_us is None else bool(span_ms * 1e3 <= round_period_us),
                optimiser_round_period_us=None if round_period_us is None else round(round_period_us, 2),
                round_span_us=round(pp['round_span_ms'] * 1e3, 2), slowest_workgroup_us=round(slow_us, 2),
                frac_slowest_workgroup=round(nbytes / (slow_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if slow_us > 0 else None,
                workgroup_busy_us=round(pp['workgroup_busy_ms'] * 1e3, 2),
                alone_per_round_us=round(alone_ms * 1e3, 2),
                frac_alone=round(nbytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                alone_note='HIP events on the launch stream around ONE resident launch serving 100 rounds from the ring, / 100',
                workgroups=pp['workgroups'], tiles_per_workgroup=pp['tiles_per_workgroup'],
                bytes_moved_per_round_algorithmic=moved,
                frac_of_peak_for_bytes_moved=round(moved / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                bytes_note='algorithmic_bytes is SURVEY 8(d)\'s per-launch figure (constants re-read every round), kept so that the '
                           'fraction is comparable with the earlier rounds; the resident pass reads the 18.2 MB of constants ONCE '
                           'per fit, so frac can exceed what the bytes actually moved per round (bytes_moved_per_round_algorithmic) '
                           'would give - the pass is latency-bound, not bandwidth-bound',
                traffic=None if table is None else table['traffic_bytes_per_round'],
                frac_of_peak_for_traffic=None if table is None else round(table['traffic_bytes_per_round'] / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                traffic_source=None if table is None else
                ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate passes collected in this run on %d '
                 'stand-alone dispatches of the resident kernel serving %d closure rounds from the ring (2 x FETCH_SIZE + WRITE_SIZE, '
                 'per round; the basis once per dispatch.  It is what reached the fabric, far below the bytes the kernel moves: the '
                 'round\'s 66 KB of operands are requested from the fabric once per XCD and served to the other 26 workgroups of that '
                 'XCD by its L2, and the vertices of consecutive rounds overwrite the same 2.6 MB buffer, whose lines the write-back '
                 'L2s merge before they leave - the pass has no streaming traffic left, it is latency-bound)'
                 % (table['dispatches'], table['rounds_served'])),
                mfma_util=None if table is None else mfma_util(table['mfma_busy_cycles_per_round'], span_ms * 1e3),
                mfma_util_note='SQ_VALU_MFMA_BUSY_CYCLES per round / (avg_launch_us x 2.4 GHz x 1024 SIMDs)',
                timed_region='every closure round of one complete fit, stamped inside the resident kernel (s_memrealtime) on the pass '
                             'stream; profiles/: rocprofv3 --kernel-trace of the same command shows the resident dispatch itself',
                launch_flavour='asynchronous fit: ONE resident launch per fit, basis stationary in registers, operands from the ring '
                               '(sc1 loads), non-temporal vertex stores, the optimiser kernel on %d other CUs' % min(B, 128),
                launches_per_fit=1, rounds_per_fit=(passes or {}).get('run'))
    return roof


def mfma_util(busy_cycles, launch_us):
    """Matrix-pipe utilisation of a launch: SQ_VALU_MFMA_BUSY_CYCLES (PMC, summed over the chip's SIMDs) over launch
    duration x 2.4 GHz x 1024 SIMDs.  Small by design: the contraction is the only MFMA work of the path (north_star)."""
    return None if busy_cycles is None else round(busy_cycles / (launch_us * 1e-6 * 2.4e9 * 1024), 4)


def build_inputs(eng, syn, frame_lo, frame_hi, persons, views, seed_base=1000, per_frame_noise=False):
    """Synthetic inputs of this rank: for every person (own fixed shape) the frames [frame_lo, frame_hi) - GT parameter
    draws (seeded by person and GLOBAL frame index) -> keypoints (by the GPU forward) -> noisy 2-D observations +
    confidences; initial parameters = zeros, scale 1.  Problem order: person-major."""
    cams = syn.make_camera_ring(views)
    nf = frame_hi - frame_lo
    B = persons * nf
    xgt = np.zeros((B, 118), np.float32)
    for p in range(persons):
        betas = None if persons == 1 else np.random.default_rng(9000 + p).normal(0, 0.5, 10)
        fr = syn.make_frames(nf, seed0=seed_base + 100000 * p + frame_lo, betas=betas)
        for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items():
            xgt[p * nf:(p + 1) * nf, a:b] = fr[k]
    eng.set_problems(cams, np.zeros((B, views, 17, 2), np.float32), np.ones((B, views, 17), np.float32))
    _, joints = eng.vertices(xgt)
    if per_frame_noise:
        # strong scaling splits ONE fixed problem set over the ranks: the observation noise is drawn per (person, GLOBAL frame),
        # so that a frame's problem does not depend on which rank's shard it falls into (tests/test_gpu_bench_8rank.py compares
        # the 8-rank result with the 1-rank result bit for bit).  The weak-scaling workloads keep the per-shard draw of the
        # earlier rounds (rank 0's frames 0..31 are the same problems at every N).
        jn = joints.cpu().numpy()
        parts = [syn.make_observations(jn[p * nf + f:p * nf + f + 1], cams, seed=seed_base + 7 + 100000 * p + frame_lo + f)
                 for p in range(persons) for f in range(nf)]
        gt, conf = np.concatenate([a for a, _ in parts]), np.concatenate([b for _, b in parts])
    else:
        gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=seed_base + frame_lo + 7)
    eng.set_problems(cams, gt, conf)
    x0 = np.zeros((B, 118), np.float32)
    x0[:, 85] = 1.0
    return cams, gt, conf, x0


def demo_inputs(syn):
    """BASELINE configs[0]: the reference's shipped demo - real cameras, keypoints, VPoser checkpoint and the
    reference's own initial guess, from the committed golden files (oracle/make_golden_demo.py)."""
    gd = os.path.join(ROOT, 'tests', 'golden')
    g = dict(np.load(os.path.join(gd, 'demo_fit_smpl.npz')))
    vpw = {k: v for k, v in np.load(os.path.join(gd, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    lsp = np.load(os.path.join(gd, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(lsp['rows'], lsp['cols'], lsp['vals']))
    cams = tuple(g[k].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    x0 = np.zeros((1, 118), np.float32)
    x0[0, 85] = 1.0
    f = g['x0']                                  # reference order with VPoser: betas go transl scale embedding
    x0[0, 0:10] = f[0:10]; x0[0, 10:13] = f[10:13]; x0[0, 82:85] = f[13:16]; x0[0, 85] = f[16]; x0[0, 86:118] = f[17:49]
    return g, vpw, model, cams, g['gt_xy'][None].astype(np.float32), g['conf'][None].astype(np.float32), x0


def demo_spread_timing(eng, g, stages, n=48):
    """The demo frame's fit time as a MEDIAN over starts: the reference's initial guess and 47 copies perturbed by 1e-6 (the first
    48 starts of tests/golden/demo_spread192.npz, the reference's own spread recording), each fitted ALONE (one problem + its
    decoder helpers, like `value`'s run) - the single trajectory of the default start moves by +-35 % in closures with the last
    bit of one gradient word, the median over 48 starts does not."""
    fn = os.path.join(ROOT, 'tests', 'golden', 'demo_spread192.npz')
    if not os.path.isfile(fn):
        return None
    x0s = np.load(fn)['x0'][:n].astype(np.float32)
    ms, ncl, fl = [], [], []
    for i in range(x0s.shape[0]):
        xd = torch.tensor(x0s[i:i + 1], device=eng.device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xf, st = eng.fit(xd, stages)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0)); ncl.append(int(st['n_closure'][0])); fl.append(float(st['final_loss'][0]))
    ms, ncl = np.asarray(ms), np.asarray(ncl)
    return dict(starts=int(x0s.shape[0]), ms_per_fit_median=round(float(np.median(ms)), 3), ms_per_fit_min=round(float(ms.min()), 3),
                ms_per_fit_max=round(float(ms.max()), 3), closures_median=int(np.median(ncl)), closures_min=int(ncl.min()),
                closures_max=int(ncl.max()), us_per_round_median=round(float(np.median(1e3 * ms / np.maximum(ncl, 1))), 2),
                final_loss_median=round(float(np.median(fl)), 1),
                note='each start fitted alone, wall clock around mvfit_fit; the starts of the reference\'s own spread recording '
                     '(oracle/make_golden_demo_spread.py)')


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline_reference(model, cams, gt, conf, stages, use_vposer, vpw, budget_s=10.0):
    """north_star: "the reference timed on the same box's host cores (core count stated) in the same run".  The reference
    itself - create_fitting_closure + LBFGSLs + run_fitting in the stage loop of non_linear_solver (float32, the frames of this
    very batch, from the same start) - imported from /root/reference (build container) or from the archive `make -C oracle
    stage` ships to the GPU box (oracle/_ref/reference_stage.tgz; test infrastructure, see oracle/ref_import.py).  Timed at 1
    thread and at 8 threads (SURVEY section 6's two settings) for ~budget_s each, plus the raw closure rate.  None when no
    reference is present."""
    from oracle import ref_import as ri
    if not ri.available():
        return None
    ncpu = os.cpu_count() or 1
    runs = {}
    for threads in (1, min(8, ncpu)):
        r = ri.time_reference_fits(model, cams, gt, conf, stages, use_vposer=use_vposer, vposer_weights=vpw, threads=threads,
                                   budget_s=budget_s)
        raw = ri.time_reference_closure(model, cams, gt[0], conf[0], stages[0], use_vposer=use_vposer, vposer_weights=vpw,
                                        threads=threads, calls=100)
        runs[threads] = dict(closures_per_s=round(r['closures'] / r['seconds'], 1), frames=r['frames'], closures=r['closures'],
                             seconds=round(r['seconds'], 2), ms_per_frame=round(1e3 * r['seconds'] / r['frames'], 1),
                             final_loss_first_frame=float(r['final_losses'][0]) if r['final_losses'][0] is not None else None,
                             raw_closure_fwd_bwd_per_s=round(raw, 1))
    torch.set_num_threads(1)
    best = max(runs, key=lambda t: runs[t]['closures_per_s'])
    return dict(value=runs[best]['closures_per_s'], unit='closures/s', cores=best, kind='reference',
                sample='%d of the %d frames of this batch, full 4-stage fits from the same start (%d closures in %.1f s): the '
                       'reference itself (code/utils/fitting.py create_fitting_closure + run_fitting, code/optimizers/lbfgs_ls.py '
                       'LBFGSLs; float32, PyTorch %s CPU), %d thread(s) - the faster of the 1- and %d-thread runs - of %d host '
                       'cores (%s); %s' % (runs[best]['frames'], gt.shape[0], runs[best]['closures'], runs[best]['seconds'],
                                           torch.__version__, best, max(runs), ncpu, cpu_model_name(),
                                           'imported from the staged archive oracle/_ref/reference_stage.tgz' if ri.STAGED
                                           else 'imported from /root/reference'),
                host_cores=ncpu, cpu_model=cpu_model_name(),
                threads={'threads_%d' % t: v for t, v in runs.items()})


def cpu_baseline(model, cams, gt, conf, stages, use_vposer, vpw, budget_s=15.0):
    """The PyTorch-CPU port of the reference closure + L-BFGS (oracle/closure_torch.py; the reference itself cannot
    travel to the GPU box): full 4-stage fits of the first frames of the same batch on ONE thread (the reference is
    fastest single-threaded) for ~budget_s; plus the raw closure (forward + backward) rate at 1 thread
    and at os.cpu_count() threads on a bounded number of calls (the all-threads run is an honest data point, not a
    recommendation: intra-op threading of these tiny tensor ops is slower than one thread)."""
    from oracle import closure_np as cn
    from oracle import closure_torch as ct
    lay, D = cn.param_layout(use_vposer)
    x0 = np.zeros(D)
    x0[lay['scale'][0]] = 1.0
    torch.set_num_threads(1)
    t0 = time.time()
    ncl = 0
    nfr = 0
    for b in range(gt.shape[0]):
        tc = ct.TorchClosure(model, cams, gt[b], conf[b], vposer=vpw)
        _, _, n = ct.fit_one(tc, x0, stages, use_vposer)
        ncl += n
        nfr += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    out = dict(value=round(ncl / dt, 1), unit='closures/s', cores=1, kind='port',
               sample='%d of the %d frames, full 4-stage fits (%d closures, %.1f s), PyTorch %s CPU port of the '
                      'reference closure + L-BFGS, 1 thread of %d host cores, %.0f ms/frame'
                      % (nfr, gt.shape[0], ncl, dt, torch.__version__, os.cpu_count(), 1e3 * dt / nfr))
    tc = ct.TorchClosure(model, cams, gt[0], conf[0], vposer=vpw)
    raw = {}
    for threads, calls in ((1, 60), (os.cpu_count() or 1, 6)):
        torch.set_num_threads(threads)
        tc.evaluate(x0, stages[0], use_vposer)                       # warm
        t1 = time.time()
        done = 0
        for _ in range(calls):
            tc.evaluate(x0, stages[0], use_vposer)
            done += 1
            if time.time() - t1 > 8.0:
                break
        raw[threads] = dict(closures_per_s=round(done / (time.time() - t1), 2), calls=done)
    torch.set_num_threads(1)
    out['raw_closure_fwd_bwd'] = {'threads_%d' % t: v for t, v in raw.items()}
    return out


def end_to_end(MvFit, syn, frames=32, reps=3):
    """The whole per-frame pipeline the reference times as one (code/main.py:27,91-94): keypoint / camera files -> initial
    guess -> 4-stage fit -> decoded pose -> result pkl, by mvsmplfitting_amd.batch.fit_folder on the shipped demo frame
    (tests/golden/demo_data: 6 real views, VPoser checkpoint) replicated to `frames` frames of one serial.  Best of `reps`."""
    import shutil
    import tempfile
    from mvsmplfitting_amd import batch
    gold = os.path.join(ROOT, 'tests', 'golden')
    data = os.path.join(gold, 'demo_data')
    vpw = {k: v for k, v in np.load(os.path.join(gold, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    lsp = np.load(os.path.join(gold, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(lsp['rows'], lsp['cols'], lsp['vals']))
    tmp = tempfile.mkdtemp(prefix='mvfit_e2e_')
    try:
        for cam in sorted(os.listdir(os.path.join(data, 'keypoints', '0000'))):
            src = os.path.join(data, 'keypoints', '0000', cam)
            fn = sorted(os.listdir(src))[0]
            dst = os.path.join(tmp, 'keypoints', '0000', cam)
            os.makedirs(dst)
            for f in range(frames):
                shutil.copyfile(os.path.join(src, fn), os.path.join(dst, '%05d_keypoints.json' % (f + 1)))
        eng = MvFit(model, vposer=vpw)
        best = None
        for rep in range(reps + 1):                       # (first pass: warm-up)
            tm = {}
            t0 = time.time()
            res = batch.fit_folder(model, os.path.join(tmp, 'keypoints'), os.path.join(data, '3DOH50K_Parameters.txt'),
                                   os.path.join(tmp, 'results_%d' % rep), vposer=vpw, engine=eng, timing=tm)
            torch.cuda.synchronize()
            wall = time.time() - t0
            if rep and (best is None or wall < best['wall_s']):
                r = res['0000']
                best = dict(wall_s=wall, steps_s=tm, closures=int(r['n_closure'].sum()), final_loss_median=float(np.median(r['final_loss'])))
        eng.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {'workload': 'batch.fit_folder: the demo frame (6 real views, shipped VPoser checkpoint) x %d frames of one serial: files -> '
                        'initial guess -> 4-stage fit -> decoded pose -> result pkl' % frames,
            'frames': frames, 'ms_per_frame_end_to_end': round(1e3 * best['wall_s'] / frames, 3),
            'ms_total': round(1e3 * best['wall_s'], 2),
            'breakdown_ms': {k: round(1e3 * v, 2) for k, v in best['steps_s'].items()},
            'closures': best['closures'], 'closures_per_s_end_to_end': round(best['closures'] / best['wall_s'], 1),
            'final_loss_median': best['final_loss_median'],
            'reference_counterpart': "code/main.py:27,91-94 (the reference's only timer: whole frame incl. file I/O); not run here"}


def vertex_pass_variants(MvFit, syn, model, views, skin_topk):
    """Roofline of the vertex pass in its other instantiations, each on its own engine: the exact-fp32 contraction
    at 32 problems and the chunk-loop kernel at 128 problems (64 back-to-back launches inside one hipEvent pair)."""
    out = {}
    cams = syn.make_camera_ring(views)

    def measure(B, opts):
        eng = MvFit(model, options=opts)
        rng = np.random.default_rng(B)
        x = np.zeros((B, 118), np.float32)
        x[:, :86] = rng.normal(0, 0.2, (B, 86))
        x[:, 85] = 1.0
        eng.set_problems(cams, np.zeros((B, views, 17, 2), np.float32), np.ones((B, views, 17), np.float32))
        eng.vertices(x)
        torch.cuda.synchronize()
        ms = min(eng.profile_vertex_pass_ms(64) for _ in range(3))
        eng.close()
        return ms
    for name, B, env, kern, key in (
            ('exact_fp32_B32', 32, {'contraction': 'exact_fp32'}, 'lbs_vertex_pass_kernel<true>', None),
            ('split_fp16_B128', 128, {}, 'lbs_vertex_pass_pipe_kernel', 'B128'),
            ('half_basis_B32', 32, {'contraction': 'half_basis'}, 'lbs_vertex_pass_split_kernel<true>', None),
            ('half_basis_B128', 128, {'contraction': 'half_basis'}, 'lbs_vertex_pass_pipe_kernel', None)):
        ms = measure(B, env)
        nbytes = bytes_fwd(B, skin_topk, half_basis=env.get('contraction') == 'half_basis')
        ach = nbytes / (ms * 1e-3) / 1e9
        if B > 32 and not (skin_topk and skin_topk <= 4):
            kern = 'lbs_vertex_pass_split_loop_kernel<false>'     # dense skinning rows: the lock-step chunk loop
        out[name] = dict(kernel=kern if skin_topk else kern.replace('<true>', '<false>'), problems=B,
                         avg_launch_us=round(ms * 1e3, 2), algorithmic_bytes=nbytes,
                         achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                         measured='the per-round launch kernel ALONE (64 back-to-back launches inside one hipEvent pair)',
                         traffic=pmc_value(key, 'vertex_pass_pipe') if key else None)
    # the pass INSIDE a fit at 128 problems (the per-GPU share of configs[3]): one complete profiled 4-stage fit
    for name, env in (('split_fp16_B128', {}), ('half_basis_B128', {'contraction': 'half_basis'})):
        try:
            from mvsmplfitting_amd.engine import stage_weights
            eng = MvFit(model, options=env)
            cams_b, gt, conf, x0 = build_inputs(eng, syn, 0, 128, 1, views)
            st = stage_weights(1536.0)
            x0_d = torch.tensor(x0, device=eng.device)
            eng.fit(x0_d, st)                                      # warm
            eng.profile(True)
            xf, stt = eng.fit(x0_d, st)
            pr = eng.profile_read()
            pp = eng.pass_profile()
            eng.profile(False)
            nbytes = bytes_fwd(128, skin_topk, half_basis=env.get('contraction') == 'half_basis')
            ms = pr['vertex_pass_ms']
            which = 'dispatch'
            if pp['tiles_per_workgroup']:
                # (resident pass: span while it keeps up with the optimiser, the slowest workgroup's service time otherwise)
                t0 = time.perf_counter()
                eng.fit(x0_d, st)
                torch.cuda.synchronize()
                period_us = 1e6 * (time.perf_counter() - t0) / max(int(stt['n_closure'].max().item()), 1)
                us, which = resident_round_us(pp, period_us)
                ms = us * 1e-3
            e = out[name]
            e['in_fit'] = dict(
                kernel=pp['kernel'] if pp['tiles_per_workgroup'] else e['kernel'],
                avg_launch_us=round(ms * 1e3, 2), avg_launch_is=which, rounds=pr['vertex_pass_launches'],
                round_span_us=round(pp['round_span_ms'] * 1e3, 2), slowest_workgroup_us=round(pp['slowest_workgroup_ms'] * 1e3, 2),
                frac=round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                frac_true=(round(true_bound(128, pp['form'], pp['workgroups'], env.get('contraction') == 'half_basis')['us'] / (ms * 1e3), 4)
                           if pp['tiles_per_workgroup'] and ms > 0 else None),
                bound_true=(true_bound(128, pp['form'], pp['workgroups'], env.get('contraction') == 'half_basis')
                            if pp['tiles_per_workgroup'] else None),
                workgroups=pp['workgroups'], workgroup_busy_us=round(pp['workgroup_busy_ms'] * 1e3, 2),
                alone_per_round_us=round(min(eng.profile_resident_pass_ms(100) for _ in range(3)) * 1e3, 2) if pp['tiles_per_workgroup'] else None,
                passes=stt['passes'],
                note='service span of a closure round inside one complete 128-problem fit (resident pass: stamped in the kernel; '
                     'per-round launches: the dispatches\' own begin / end)')
            if e['in_fit']['frac'] is not None:
                e['frac_in_fit'] = e['in_fit']['frac']
            eng.close()
        except Exception as ex:                                 # noqa: BLE001 - an extra must not kill the headline line
            out[name]['in_fit'] = {'error': repr(ex)}
    return out


def prior_variants(MvFit, syn, _lib, stage_weights, model, frames, views, steps=2):
    """The same workload with the other pose priors (VPoser-L2 - the reference's shipped yaml default,
    use_vposer: true - and GMM): whole-fit closures/s, one warm-up fit + `steps` timed fits each, own engine."""
    out = {}
    for name, flag, kw in (('prior_vposer', _lib.F_VPOSER, dict(vposer=syn.make_vposer_decoder())),
                           ('prior_gmm', _lib.F_PRIOR_GMM, dict(gmm=syn.gmm_constants(syn.make_gmm())))):
        eng = MvFit(model, **kw)
        cams, gt, conf, x0 = build_inputs(eng, syn, 0, frames, 1, views)
        stages = stage_weights(1536.0, flags=flag)
        x0_d = torch.tensor(x0, device=eng.device)
        eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ncl = []
        for _ in range(steps):
            xf, st = eng.fit(x0_d, stages)
            ncl.append(st['n_closure'])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tot = sum(int(n.sum().item()) for n in ncl)
        out[name] = dict(value=round(tot / dt, 1), unit='closures/s', ms_per_step=round(1e3 * dt / steps, 3),
                         closures_per_fit_per_frame=round(tot / steps / frames, 1), closure_rounds_per_fit=int(ncl[-1].max().item()),
                         final_loss_median=float(np.median(st['final_loss'].cpu().numpy())), vertex_passes_last_fit=st.get('passes'),
                         vertex_pass_mode=eng.pass_profile())
        if flag & _lib.F_VPOSER:
            out[name]['decoder_helpers_last_fit'] = eng.decoder_stats()
        eng.close()
    return out


def reuse_variant(eng, _lib, stage_weights, x0_d, steps=3):
    """Opt-in MVFIT_F_REUSE_OUTER_VALUE on the headline workload (same engine, same inputs): LBFGS.step() opens with a
    closure call at a point whose value the optimiser still holds (8-10 % of the reference's closure calls); with the
    flag the device feeds it back instead of evaluating again.  Same iterates - the final losses must be identical -,
    fewer closures: a time-to-solution number, not the headline closures/s."""
    out = {}
    for name, flags in (('reference_closure_count', 0), ('reuse_outer_value', _lib.F_REUSE_OUTER_VALUE)):
        stages = stage_weights(1536.0, flags=flags)
        eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            xf, st = eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = dict(ms_per_step=round(1e3 * dt / steps, 3), closures_per_fit=int(st['n_closure'].sum().item()),
                         closure_rounds_per_fit=int(st['n_closure'].max().item()),
                         final_loss_sum=float(st['final_loss'].double().sum().item()))
    out['identical_final_losses'] = out['reference_closure_count']['final_loss_sum'] == out['reuse_outer_value']['final_loss_sum']
    return out


def sdf_faces_variants(eng, model, stages, x0_d, grid, steps=2):
    """the interpenetration term as the reference wires it (faces.reshape(1,-1,3): the op sees ONE
    triangle, fitting.py:367-368) and on all 13,776 faces (what sdf_cuda_kernel.cu:258-287 does with a proper face list) - the
    latter on per-round face lists (sdf_term.hip), bit-identical to the walk over every face (tests/test_gpu_sdf_cull.py).
    With all faces every surface vertex has inside corners, the stage-3/4 penalty is ~1e8 and flat along the search
    directions, and the strong-Wolfe bracketing phase extrapolates the parameters out to infinity (final losses NaN): the
    figure is the cost of the term, not a usable fit."""
    out = {}
    for name, nf in (('as_wired_one_triangle', 1), ('all_faces', None)):
        eng.set_sdf(model['faces'], num_faces=nf, grid_size=grid)
        eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            xf, st = eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fl = st['final_loss'].cpu().numpy()
        out[name] = dict(ms_per_step=round(1e3 * dt / steps, 3), closures_per_fit=int(st['n_closure'].sum().item()),
                         closure_rounds_per_fit=int(st['n_closure'].max().item()),
                         closures_per_s=round(int(st['n_closure'].sum().item()) * steps / dt, 1),
                         problems_with_finite_final_loss=int(np.isfinite(fl).sum()))
    eng.set_sdf(model['faces'], num_faces=1, grid_size=grid)
    return out


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='configs1', choices=['configs1', 'demo', 'configs0', 'configs2', 'configs3', 'configs4', 'folder'],
                    help='named BASELINE.json workload (see the module docstring); --frames / --views / --persons override a preset')
    ap.add_argument('--frames', type=int, default=None, help='frames per GPU and person (with --strong: the total)')
    ap.add_argument('--views', type=int, default=None)
    ap.add_argument('--persons', type=int, default=None, help='subjects (own shape each); every GPU fits its frames of all of them')
    ap.add_argument('--prior', default='l2', choices=['l2', 'vposer', 'gmm'])
    ap.add_argument('--sparse', action='store_true',
                    help='objective-vertices-only closure (no full vertex pass per closure)')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --frames is the TOTAL, split over the ranks (default: weak, --frames per GPU)')
    ap.add_argument('--skin-topk', type=int, default=4,
                    help='non-zero skinning weights per vertex of the synthetic body (SMPL: <= 4); 0 = dense rows')
    ap.add_argument('--sdf', action='store_true',
                    help='configs[2]: SDF interpenetration term on (grid 128; yaml coll_loss_weights)')
    ap.add_argument('--sdf-faces', default='wired', choices=['wired', 'all'],
                    help="wired: the first triangle only, as the reference's call site makes the op see it; all: the 13,776 faces")
    ap.add_argument('--contraction', default='split_fp16', choices=['split_fp16', 'exact_fp32', 'half_basis'],
                    help='blendshape contraction of the vertex pass (mvfit_options::contraction)')
    ap.add_argument('--round-mode', default='auto', choices=['auto', 'chained'],
                    help='chained: vertex pass -> step kernel per closure round also without the SDF term (mvfit_options::round_mode)')
    ap.add_argument('--resident-pass', type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help='vertex passes of the asynchronous fit: -1 automatic, 0 per-round launches, 1 / 2 / 3 resident forms (mvfit_options.resident_pass)')
    ap.add_argument('--vposer-sets', type=int, default=0, help='decoder-helper sets of a VPoser fit (0 automatic; mvfit_options::vposer_sets)')
    ap.add_argument('--work-queue', type=int, default=1, choices=[0, 1],
                    help='more problems than optimiser workgroups: 1 one launch with a work queue (default), 0 sub-batches (mvfit_options.work_queue)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='skip the extra vertex-pass roofline / prior variants')
    ap.add_argument('--no-pmc', action='store_true', help='do not run the rocprofv3 PMC passes (traffic then comes from profiles/*_pmc.json)')
    ap.add_argument('--dist-backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for a dry run)')
    ap.add_argument('--dump-gathered', default=None,
                    help='rank 0 writes the gathered parameters of the last fit, [persons, frames_total, 118] float32, to this .npy '
                         '(tests compare an N-rank run with a 1-rank run bit for bit)')
    ap.add_argument('--single-device', action='store_true',
                    help='dry run of the multi-rank path on ONE GPU: every rank uses cuda:0 (needs --dist-backend gloo)')
    args = ap.parse_args()

    # ---- --gpus N without a launcher: start the N ranks here (one process per GPU, torch.distributed over 127.0.0.1) ----
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    from mvsmplfitting_amd import _lib
    from mvsmplfitting_amd import synthetic as syn
    from mvsmplfitting_amd.engine import MvFit, stage_weights
    from mvsmplfitting_amd.sharding import gather_results, shard_range

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    if args.config == 'folder':
        assert world == 1, '--config folder is a single-GPU line'
        e2e = end_to_end(MvFit, syn, frames=args.frames or 32, reps=max(1, args.steps))
        print(json.dumps({'metric': 'end-to-end ms per frame (keypoint files -> initial guess -> 4-stage fit -> result files)',
                          'value': e2e['ms_per_frame_end_to_end'], 'unit': 'ms/frame', 'n_gpus': 1, 'steps': args.steps,
                          'warmup': 1, 'ms_per_step': e2e['ms_total'], 'higher_is_better': False, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'the reference\'s shipped demo frame, replicated',
                          'config': {'workload': e2e['workload']}, 'end_to_end': e2e}))
        return
    demo = args.config in ('demo', 'configs0')
    preset = dict(PRESETS.get(args.config, PRESETS['configs1']))
    frames = args.frames if args.frames is not None else preset['frames']
    views = args.views if args.views is not None else preset['views']
    persons = args.persons if args.persons is not None else preset['persons']
    sdf = args.sdf or preset.get('sdf', False)
    half_basis = bool(preset.get('half_basis')) or args.contraction == 'half_basis'
    if demo:
        assert world == 1, 'the demo is one frame'
        g, vpw, model, cams, gt, conf, x0 = demo_inputs(syn)
        args.prior, views, args.skin_topk, persons = 'vposer', 6, 0, 1
        gmm = None
        B = total = total_frames = 1
        lo = 0
    else:
        model = syn.make_body_model(0, skin_topk=args.skin_topk or None)
        vpw = syn.make_vposer_decoder() if args.prior == 'vposer' else None
        gmm = syn.make_gmm() if args.prior == 'gmm' else None
        total_frames = frames if args.strong else frames * world
        lo, hi = shard_range(total_frames, world, rank)       # contiguous global frames of this rank; seeds follow the global index
        assert hi > lo, 'more ranks than frames'
        B = persons * (hi - lo)
        total = persons * total_frames
    eng_options = dict(contraction='half_basis' if half_basis else args.contraction, round_mode=1 if args.round_mode == 'chained' else 0,
                       resident_pass=args.resident_pass, vposer_sets=args.vposer_sets, work_queue=args.work_queue)
    if args.single_device and world > 1 and args.resident_pass == -1:
        # the dry run of the N-rank path puts N processes on ONE GPU: the resident pass assumes that a fit's workgroups (the
        # optimiser's + the pass's, ~250 CUs) are resident together, which several processes sharing the device cannot all have
        # (they would wait for one another's CUs until the ring's 20 ms patience ends; include/mvfit.h: resident_pass) - the dry
        # run uses the per-round pass launches; one process per GPU, the deployment, keeps the automatic choice
        eng_options['resident_pass'] = 0
    eng = MvFit(model, vposer=vpw, gmm=None if gmm is None else syn.gmm_constants(gmm), device=local_rank, options=eng_options)
    flags = 0
    if args.prior == 'vposer':
        flags |= _lib.F_VPOSER
    if args.prior == 'gmm':
        flags |= _lib.F_PRIOR_GMM
    if args.sparse:
        flags |= _lib.F_SPARSE_VERTS
    stages = stage_weights(1536.0, flags=flags, coll_w=[0.0, 0.0, 1000.0, 4500.0] if sdf else None)
    if sdf:                                                  # fit_smpl.yaml:55-59, fitting.py:367-368
        eng.set_sdf(model['faces'], num_faces=1 if args.sdf_faces == 'wired' else None, grid_size=128)
    if demo:
        eng.set_problems(cams, gt, conf)
    else:
        cams, gt, conf, x0 = build_inputs(eng, syn, lo, hi, persons, views, per_frame_noise=args.strong)
    x0_d = torch.tensor(x0, device=dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def gather(x):
        # per-person blocks of this rank's frames -> [persons, frames_total, 118]: one all_gather (the path's only collective)
        nf = x.shape[0] // persons
        return gather_results(x.reshape(persons, nf, -1).transpose(0, 1).contiguous(), total_frames)

    for _ in range(max(args.warmup, 1) if args.warmup else 0):
        xw, st = eng.fit(x0_d, stages)
        if world > 1:
            gather(xw)                            # also sets up the RCCL rings outside the timed region
    # torch loads its own reduction / copy kernels lazily on first use (~80 ms): touch the exact ops of
    # the timed loop once here (also with --warmup 0), so that module loading is not billed to a fit
    _z = torch.zeros(B, device=dev, dtype=torch.int32)
    int(_z.sum().item()); int(_z.max().item())
    barrier()
    t0 = time.perf_counter()
    n_closure = 0
    n_iter = 0
    finals = None
    gathered = None
    counts = []
    passes = None
    lost = [0, 0]
    busy = 0.0
    for _ in range(args.steps):
        tb = time.perf_counter()
        xf, st = eng.fit(x0_d, stages)             # returns when this rank's GPU has finished the fit
        busy += time.perf_counter() - tb
        if world > 1:
            gathered = gather(xf)                  # RCCL over xGMI, inside the step
        counts.append((st['n_closure'], st['n_iter']))      # device tensors: read after the timed region
        finals = st['final_loss']
        passes = st.get('passes')
        if passes:
            lost[0] += passes['missed']; lost[1] += passes['timed_out']
    barrier()
    dt = time.perf_counter() - t0
    # VPoser prior: the decoder layers of the single-launch fits run on helper workgroups (csrc/vposer_service.h)
    decoder = eng.decoder_stats() if (flags & _lib.F_VPOSER) else None
    for ncl_t, nit_t in counts:
        n_closure += int(ncl_t.sum().item())
        n_iter += int(nit_t.sum().item())
    n_max = int(counts[-1][0].max().item())

    # max-over-ranks time, totals, per-rank busy time, the form the vertex passes took on every rank
    tot_closure, tot_iter, tmax = n_closure, n_iter, dt
    busy_all = [round(1e3 * busy / args.steps, 3)]
    form_all = [int(eng.pass_profile()['form'])] if not (args.sparse or sdf) else [None]
    if world > 1:
        import torch.distributed as dist
        red = torch.tensor([float(n_closure), float(n_iter), float(lost[0]), float(lost[1])], device=dev, dtype=torch.float64)
        dist.all_reduce(red)
        tm = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        bz = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(bz, torch.tensor([busy / args.steps * 1e3], device=dev, dtype=torch.float64))
        busy_all = [round(float(b.item()), 3) for b in bz]
        if form_all[0] is not None:
            fz = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(fz, torch.tensor([float(form_all[0])], device=dev, dtype=torch.float64))
            form_all = [int(f.item()) for f in fz]
        assert gathered.shape[0] == total_frames
        tot_closure, tot_iter, tmax = int(red[0].item()), int(red[1].item()), float(tm.item())
        lost = [int(red[2].item()), int(red[3].item())]
    if args.dump_gathered and rank == 0:
        # [frames_total, persons, 118] (the gather's layout: contiguous frame blocks per rank) -> [persons, frames_total, 118]
        full = gathered if world > 1 else xf.reshape(persons, B // persons, -1).transpose(0, 1)
        np.save(args.dump_gathered, full.transpose(0, 1).contiguous().cpu().numpy())

    # roofline of the dominant HBM kernel (LBS vertex pass), measured live with HIP events on the stream the kernel is
    # launched on
    roof = None
    if not args.sparse and not demo:
        in_fit = bool(passes and passes['run'] > 0)          # asynchronous fit: ring operands, non-temporal streams
        nchunks = (B + 31) // 32
        sparse_w = bool(args.skin_topk and args.skin_topk <= 4)
        if nchunks == 1:
            kname = 'lbs_vertex_pass_split_kernel' + ('<true>' if sparse_w else '<false>')
        else:        # more than 32 problems: the two-role pipelined chunk loop (<= 4 weights per vertex), else the lock-step loop
            kname = 'lbs_vertex_pass_pipe_kernel' if sparse_w else 'lbs_vertex_pass_split_loop_kernel<false>'
        # (a) the kernel as it ran inside a fit: one more identical fit (outside the timed region) in which every pass
        #     launch carries an event pair stamped by the runtime with the dispatch's own begin / end (hipExtLaunchKernelGGL)
        eng.profile(True)
        eng.fit(x0_d, stages)
        pr = eng.profile_read()
        pp = eng.pass_profile()
        eng.profile(False)
        # (b) the kernel alone: 64 back-to-back launches inside one hipEvent pair, in the fit's launch flavour and with
        #     plain loads
        resident = in_fit and pp['tiles_per_workgroup'] > 0 and pp['rounds_stamped'] > 0
        # (more than 128 problems are fitted in sub-batches: the resident pass serves one sub-batch's problems per round)
        B_pass = B if not resident or B <= 128 else -(-(-(-B // -(-B // 128))) // 32) * 32
        nbytes = bytes_fwd(B_pass, args.skin_topk, half_basis)
        standard = views == 8 and args.skin_topk == 4 and not half_basis
        if resident:
            roof = resident_roofline(eng, B_pass, nbytes, pr, pp, passes,
                                     with_pmc=rank == 0 and world == 1 and not args.no_pmc and standard,
                                     round_period_us=1e6 * tmax / args.steps / max(n_max, 1) / max(-(-B // B_pass), 1),
                                     half_basis=half_basis)
        ms_b2b = min(eng.profile_vertex_pass_ms(64, as_in_async_fit=in_fit) for _ in range(3))
        ms_plain = min(eng.profile_vertex_pass_ms(64) for _ in range(3))
        ms_fit = pr['vertex_pass_ms'] if pr['vertex_pass_launches'] > 0 else ms_b2b
        ach = nbytes / (ms_fit * 1e-3) / 1e9
        # (c) HBM-side traffic / matrix-pipe cycles per launch: rocprofv3 PMC passes run NOW on this GPU (rank 0, one GPU),
        #     else the committed summary of the same kernel
        pmc_key = 'B%d' % B
        table, src = None, None
        if rank == 0 and world == 1 and not args.no_pmc and standard and not resident:
            table = measure_pmc(B)
            src = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate passes, collected in this run (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 wide-read correction)'
        if table is None and standard and pmc_file():
            src = '%s (rocprofv3 --pmc passes of an earlier run of the same kernel; tools/collect_profiles.sh)' % os.path.relpath(pmc_file(), ROOT)
        traffic = pmc_value(pmc_key, kname.split('<')[0], 'traffic_bytes', table) if standard else None
        busy_cy = pmc_value(pmc_key, kname.split('<')[0], 'mfma_busy_cycles_per_launch', table) if standard else None
        per_round = dict(bound='hbm', kernel=kname, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=src if traffic is not None else None,
                    algorithmic_bytes=nbytes, avg_launch_us=round(ms_fit * 1e3, 2),
                    mfma_util=mfma_util(busy_cy, ms_fit * 1e3),
                    mfma_util_note='SQ_VALU_MFMA_BUSY_CYCLES per launch / (avg_launch_us x 2.4 GHz x 1024 SIMDs)',
                    timed_region='every vertex-pass launch of one complete fit (%d launches), begin / end of each dispatch stamped by '
                                 'the runtime (hipExtLaunchKernelGGL events on the stream the kernel is launched on)' % pr['vertex_pass_launches'],
                    launch_flavour='asynchronous fit: operands from the ring, non-temporal basis stream and vertex stores, the optimiser '
                                   'kernel running concurrently on %d other CUs' % min(B, 160) if in_fit else 'chained mode (plain loads, side outputs)',
                    alone_back_to_back_us=round(ms_b2b * 1e3, 2), alone_back_to_back_plain_loads_us=round(ms_plain * 1e3, 2),
                    frac_alone_plain_loads=round(nbytes / (ms_plain * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    launches_per_fit=(passes or {}).get('run'))
        if resident:
            # the per-round launch kernels stay in the library (dense skinning rows, exact-fp32 contraction, launches that
            # leave no CUs): their stand-alone figures next to the resident pass
            roof['per_round_launch_kernel_alone'] = dict(kernel=kname, alone_back_to_back_us=round(ms_b2b * 1e3, 2),
                                                         alone_back_to_back_plain_loads_us=round(ms_plain * 1e3, 2),
                                                         frac_alone_plain_loads=round(nbytes / (ms_plain * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        else:
            roof = per_round
        if rank == 0 and not args.no_variants and world == 1 and args.config == 'configs1':
            roof['variants'] = vertex_pass_variants(MvFit, syn, model, views, args.skin_topk)

    if rank == 0:
        variants = None
        if not args.no_variants and world == 1 and args.config == 'configs1' and args.prior == 'l2' and not sdf and not args.sparse:
            variants = prior_variants(MvFit, syn, _lib, stage_weights, model, frames, views)
            variants['time_to_solution_opt_in'] = reuse_variant(eng, _lib, stage_weights, x0_d)
            try:
                variants['end_to_end'] = end_to_end(MvFit, syn, frames=32, reps=2)
            except Exception as e:                          # noqa: BLE001 - the headline line must not die on the extra
                variants['end_to_end'] = {'error': repr(e)}
        if not args.no_variants and world == 1 and sdf and args.sdf_faces == 'wired' and not demo:
            variants = dict(variants or {})
            variants['sdf_faces'] = sdf_faces_variants(eng, model, stages, x0_d, 128)
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.prior != 'gmm':
            cpu_stages = [dict(s) for s in stages]
            port = cpu_baseline(model, cams, gt, conf, cpu_stages, args.prior == 'vposer', vpw, budget_s=8.0)
            cpu = cpu_baseline_reference(model, cams, gt, conf, cpu_stages, args.prior == 'vposer', vpw)
            if cpu is None:                                  # no reference on this box: the port stands in, and says so
                cpu = port
            else:
                cpu['port'] = port
            cpu['reference_in_survey_container'] = dict(
                value=list(SURVEY_REFERENCE_CLOSURES_PER_S), unit='closures/s',
                note='the imported reference itself (create_fitting_closure + LBFGSLs + run_fitting, PyTorch CPU, 1 and 8 '
                     'threads of 8 vCPUs) as timed in SURVEY.md section 6; it does not exist on the GPU box')
        fl = finals.cpu().numpy()
        per_gpu = '%s%d views x %d synthetic frames%s' % ('%d persons x ' % persons if persons > 1 else '1 person x ', views,
                                                          total_frames if args.strong else (hi - lo if not demo else 1),
                                                          ' in total' if args.strong else ' per GPU')
        if demo:
            workload = ('configs[0]: the reference demo (cfg_files/fit_smpl.yaml): 1 frame x 6 real views x 1 person, real keypoints / '
                        'cameras, VPoser decoder of the shipped checkpoint, synthetic body, 4 yaml stages')
        else:
            tag = {'configs1': 'configs[1]', 'configs2': 'configs[2]', 'configs3': 'configs[3] (4 persons x 8 views x 256 frames over 8 GPUs = 4 x 32 frames per GPU)',
                   'configs4': 'configs[4] (16-view rig x 1024 frames over 8 GPUs = 128 frames per GPU, half-width blendshape operands)'}[args.config]
            if sdf and args.config == 'configs1':
                tag = 'configs[2]'
            sdf_txt = 'no SDF' if not sdf else ('SDF term, grid 128, yaml coll_loss_weights, %s' %
                                               ('first triangle only as the reference wires it' if args.sdf_faces == 'wired' else 'all 13,776 faces'))
            workload = '%s: %s, GMoF + pose prior (%s) + shape + angle priors, %s, 4 yaml stages' % (tag, per_gpu, args.prior, sdf_txt)
        if args.sparse:
            mode = 'objective-vertices-only'
        elif passes and passes['run'] > 0:
            mode = ('full 6890-vertex pass per closure, asynchronous: one optimiser kernel publishes the operands of every trial '
                    'point, the pass of closure round r runs concurrently on the other CUs (ring with back-pressure: no pass is lost)')
        else:
            mode = 'full 6890-vertex pass per closure, chained (pass -> step kernel per round)'
        out = {
            'metric': 'L-BFGS closure evaluations per second (fwd+bwd, all concurrently fitted problems), '
                      '%d-view 1-person 4-stage fits' % views,
            'value': round(tot_closure / tmax, 1), 'unit': 'closures/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * tmax / args.steps, 3),
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
            'scaling_note': ('strong: %d frames in total split over the ranks - a fit lasts as long as its slowest problem\'s serial '
                             'chain of closure rounds whatever the problem count, so the curve is bounded by the round latency, not '
                             'by the GPU count' % total_frames) if args.strong else
                            ('weak: %d frames per GPU, independent problems, no data-path collective (one final gather): linear by '
                             'construction up to the spread of the slowest problem per rank' % (B // max(persons, 1))),
            'vs_baseline': None, 'dtype': 'f32',
            'dtype_note': 'all arithmetic fp32 (line-search scalars fp64 like the reference); the blendshape contraction of the '
                          'vertex pass takes its fp32 products as error-compensated split-fp16 pairs on the matrix pipe with '
                          'fp32 accumulation (vertices 5e-7 from the float64 oracle, as with the exact fp32 chain)' +
                          ('; configs[4]: only the fp16 hi halves of the basis are streamed (vertices <= 2e-5)' if half_basis else ''),
            'data': ('real demo keypoints / cameras / VPoser checkpoint of the reference, seeded SMPL-shaped body' if demo else
                     'synthetic (seeded SMPL-shaped body%s, camera ring, noisy projected keypoints)'
                     % ((', %d skinning weights per vertex like SMPL' % args.skin_topk) if args.skin_topk else ', dense skinning rows')),
            'config': {'workload': workload, 'problems_per_gpu': B, 'frames_per_gpu': B // persons, 'persons': persons,
                       'frames_total': total_frames, 'problems_total': total, 'views': views, 'prior': args.prior,
                       'closure_mode': mode, 'parallelism': 'frame-sharded x%d, RCCL all_gather of results' % world},
            # BASELINE's "ms to convergence / frame", both readings: amortised over the concurrently fitted frames (throughput), and
            # the latency a frame actually sees (its batch's whole staged fit; one frame alone: --config demo)
            'ms_per_frame_amortised': round(1e3 * tmax / args.steps / max(B, 1), 4),
            'fit_latency_ms': round(1e3 * tmax / args.steps, 3),
            'lbfgs_iters_per_s': round(tot_iter / tmax, 1),
            'closures_per_fit_per_frame': round(tot_closure / args.steps / total, 1),
            'closure_rounds_per_fit': n_max,      # = closures of the slowest frame of rank 0's batch
            # the figure that compares builds (last review, item 4): time of one closure round of the batch = ms_per_step / rounds of
            # the slowest problem.  closures/s of a mode moves with the LENGTH of that problem's chaotic trajectory (VPoser: +-12 %
            # with the last bit of one gradient word); the time per round does not
            'us_per_round': {args.prior: round(1e3 * (1e3 * tmax / args.steps) / max(n_max, 1), 2)},
            'final_loss_median': float(np.median(fl)),
            'vertex_passes_last_fit': passes,
            'vertex_passes_lost_in_timed_fits': {'missed': lost[0], 'timed_out': lost[1]},
            'decoder_helpers_last_fit': decoder,
            'per_rank_busy_ms_per_step': busy_all,
            # form of the vertex passes on every rank (mvfit_options::resident_pass: 1 / 3 = the resident pass, one launch per fit;
            # 0 = a gate + a pass launch per closure round - expected only under --single-device, --resident-pass 0, dense skinning
            # rows or the exact-fp32 contraction)
            'per_rank_resident_form': form_all,
            # ranks that exchanged the results over RCCL in this run (0: one rank, or a gloo dry run) - no curve beyond one GPU
            # has been measured on hardware by the build sessions (one GPU per gpurun box)
            'rccl_ranks_seen': world if (world > 1 and args.dist_backend == 'nccl') else 0,
            'dist_backend': args.dist_backend if world > 1 else None,
            # the same workload with the other pose priors of SURVEY 8(d) config 2, first-class next to `value` (which is the
            # plain L2 prior, fit_smpl.yaml's body_prior_type with use_vposer off): VPoser-L2 is the reference's shipped yaml
            # default (use_vposer: true, cfg_files/fit_smpl.yaml:35-37), GMM its max-mixture prior (synthetic, M = 8)
            'value_vposer': (variants or {}).get('prior_vposer', {}).get('value') if args.prior == 'l2' else None,
            'value_gmm': (variants or {}).get('prior_gmm', {}).get('value') if args.prior == 'l2' else None,
            'variants': variants,
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if lost[0] or lost[1]:
            out['invalid_reason'] = ('%d vertex passes lost their operands / %d waits given up in the timed fits (summed over the ranks): '
                                     '"a full pass per closure" does not hold for this run' % (lost[0], lost[1]))
        expect_resident = (args.resident_pass != 0 and not (args.single_device and world > 1) and bool(args.skin_topk) and
                           args.skin_topk <= 4 and args.contraction != 'exact_fp32' and form_all[0] is not None)
        if form_all[0] is not None and (len(set(form_all)) > 1 or (expect_resident and 0 in form_all)):
            # never a quiet fallback: a rank that ran its passes as per-round launches (its fit did not get the CUs the resident
            # pass needs - a shared device, a CU mask - or an earlier fit on its ctx timed out) measures another code path
            out['invalid_reason'] = (out.get('invalid_reason', '') + ' ' if out.get('invalid_reason') else '') + \
                ('ranks ran different vertex-pass forms or fell back to per-round launches: per_rank_resident_form = %s' % form_all)
        for k_, v_ in (('vposer', 'prior_vposer'), ('gmm', 'prior_gmm')):
            e_ = (variants or {}).get(v_)
            if e_ and args.prior == 'l2':
                out['us_per_round'][k_] = round(1e3 * e_['ms_per_step'] / max(e_['closure_rounds_per_fit'], 1), 2)
        if demo:
            out['demo_48_starts'] = demo_spread_timing(eng, g, stages)
            out['reference_fit'] = dict(final_loss_fp32=float(g['fit_final32']), final_loss_fp64=float(g['fit_final64']),
                                        closures_fp32=int(g['fit_ncl32'].sum()), closures_fp64=int(g['fit_ncl64'].sum()),
                                        final_loss_spread_fp32=[float(v) for v in g['fit_spread32']],
                                        note='the reference itself on these inputs in the build container (oracle/make_golden_demo.py)')
        if os.environ.get('MVFIT_SDF_STATS_REPORT'):            # developer: work counters of the all-faces term (libmvfit_sdfstats.so)
            import ctypes as C
            st = (C.c_ulonglong * 8)()
            eng._lib.mvfit_debug_sdf_stats(st, 0)
            print('sdf stats', [int(v) for v in st], file=sys.stderr)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
